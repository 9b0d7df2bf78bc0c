// Reader for the raw arrays the job keeps on disk: hickle files (temp/raw/*.hkl, src/download_and_predict_job.py:462-463,
// :592-633 write them with hkl.dump(arr, path, mode='w', compression='gzip'), :684-714 load them with hkl.load).  A hickle file
// is an HDF5 file whose root group holds one dataset per dumped object ("data" in hickle 4/5, "data_0" in hickle 3) -- for a
// numpy array: a chunked, deflate-compressed dataset as h5py writes it with its default (earliest) format:
//     superblock v0/v1 -> root group object header (v1) -> symbol-table message -> group B-tree (v1, type 0) + local heap
//     -> symbol-table nodes -> dataset object header (v1): dataspace, datatype, data layout v3 (contiguous or chunked, with a
//     chunk B-tree v1, type 1), filter pipeline v1 (deflate = 1, shuffle = 2)
// This file parses exactly that subset of the HDF5 file format (HDF5 File Format Specification 2.0/3.0: III.A superblock,
// III.A.1 B-trees v1, III.C symbol-table nodes, III.D local heaps, IV.A.1.a object header v1, IV.A.2 messages) on the host,
// inflating chunks with zlib.  PINNED (round 4): tests/golden/hkl/*.hkl are written by the real h5py 3.3.0 / libhdf5 1.10.6
// in the layouts hickle 3.4, 4 and 5 produce (tools/gen_golden_hkl.py, run with /opt/conda/bin/python3.9: root arrays, container
// groups, attributes, gzip with and without shuffle, contiguous int64 date lists).  hickle itself is not installed anywhere in the
// image; tools/write_hdf5_fixture.py (a byte-level writer from the same specification) is kept for the malformed-file tests.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <exception>
#include <mutex>
#include <thread>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "ttc_internal.h"

namespace {

constexpr uint64_t kUndef = ~0ull;

// the file, memory-mapped read-only: nothing is copied (the query call and the read call of a caller each used to fread the whole
// file -- 130 MB of memcpy per tile), inflate reads the compressed chunks straight from the page cache
struct Bytes {
    const uint8_t* p = nullptr;
    size_t n = 0;
    const uint8_t& operator[](size_t i) const { return p[i]; }
    const uint8_t* data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
};
struct File {
    Bytes b;
    std::vector<uint8_t> owned;              // TTC_HKL_READ=1: the file's bytes, read() instead of mapped
    File() = default;
    File(const File&) = delete;
    File& operator=(const File&) = delete;
    ~File() { if (b.p && owned.empty()) munmap(const_cast<uint8_t*>(b.p), b.n); }
    bool ok(uint64_t off, uint64_t n) const { return off <= b.size() && n <= b.size() - off; }
    uint64_t u(uint64_t off, int n) const {          // little-endian unsigned of n bytes
        uint64_t v = 0;
        for (int i = 0; i < n; ++i) v |= (uint64_t)b[off + i] << (8 * i);
        return v;
    }
};

struct Msg { int type; uint64_t off; uint64_t size; };

// object header v1 (IV.A.1.a) including continuation blocks (message 0x0010)
bool read_header(const File& f, uint64_t addr, std::vector<Msg>& out, std::string& err) {
    if (!f.ok(addr, 16) || f.b[addr] != 1) { err = "unsupported object header (only version 1: file written with libver='latest'?)"; return false; }
    const int nmsg = (int)f.u(addr + 2, 2);
    const uint64_t hsize = f.u(addr + 8, 4);
    std::vector<std::pair<uint64_t, uint64_t>> blocks{{addr + 16, hsize}};
    int seen = 0;
    for (size_t bi = 0; bi < blocks.size() && seen < nmsg; ++bi) {
        uint64_t p = blocks[bi].first;
        const uint64_t end = p + blocks[bi].second;
        if (!f.ok(p, blocks[bi].second)) { err = "object header block past the end of the file"; return false; }
        while (p + 8 <= end && seen < nmsg) {
            const int type = (int)f.u(p, 2);
            const uint64_t size = f.u(p + 2, 2);
            if (p + 8 + size > end) { err = "object header message overruns its block"; return false; }
            out.push_back({type, p + 8, size});
            ++seen;
            if (type == 0x10) blocks.push_back({f.u(p + 8, 8), f.u(p + 16, 8)});
            p += 8 + size;
        }
    }
    return true;
}

// all (name, object header address) links of an old-style group: B-tree v1 type 0 -> SNOD leaves, names in the local heap
bool list_group(const File& f, uint64_t btree, uint64_t heap, std::vector<std::pair<std::string, uint64_t>>& out, std::string& err) {
    if (!f.ok(heap, 32) || std::memcmp(&f.b[heap], "HEAP", 4) != 0) { err = "bad local heap"; return false; }
    const uint64_t hdata = f.u(heap + 24, 8), hsize = f.u(heap + 8, 8);
    std::vector<uint64_t> stack{btree};
    while (!stack.empty()) {
        const uint64_t a = stack.back();
        stack.pop_back();
        if (!f.ok(a, 8)) { err = "group node past the end of the file"; return false; }
        if (std::memcmp(&f.b[a], "TREE", 4) == 0) {
            if (f.b[a + 4] != 0) { err = "group B-tree of the wrong type"; return false; }
            const int n = (int)f.u(a + 6, 2);
            uint64_t p = a + 24;                                 // key0, child0, key1, ...
            if (!f.ok(a, 24 + 16ull * n + 8)) { err = "group B-tree node past the end of the file"; return false; }
            if (stack.size() + n > 1u << 20) { err = "group B-tree too large (cyclic?)"; return false; }
            for (int i = 0; i < n; ++i) { stack.push_back(f.u(p + 8, 8)); p += 16; }
        } else if (std::memcmp(&f.b[a], "SNOD", 4) == 0) {
            const int n = (int)f.u(a + 6, 2);
            for (int i = 0; i < n; ++i) {
                const uint64_t e = a + 8 + 40ull * i;
                if (!f.ok(e, 40)) { err = "symbol table node past the end of the file"; return false; }
                const uint64_t noff = f.u(e, 8), oh = f.u(e + 8, 8);
                if (noff >= hsize || !f.ok(hdata + noff, 1)) { err = "link name outside the heap"; return false; }
                out.push_back({std::string(reinterpret_cast<const char*>(&f.b[hdata + noff])), oh});
            }
        } else { err = "unknown group node signature"; return false; }
    }
    return true;
}

struct Dataset {
    int rank = 0;
    uint64_t dims[8] = {0};
    int esize = 0, tclass = 0, is_signed = 0;
    int layout = -1;                   // 1 contiguous, 2 chunked
    uint64_t data_addr = kUndef, data_size = 0;
    uint32_t chunk[9] = {0};
    int chunk_nd = 0;                  // dimensionality in the layout message (rank + 1: the last entry is the element size)
    bool deflate = false, shuffle = false;
};

bool parse_dataset(const File& f, uint64_t oh, Dataset& d, std::string& err) {
    std::vector<Msg> msgs;
    if (!read_header(f, oh, msgs, err)) return false;
    bool have_space = false, have_type = false;
    for (const Msg& m : msgs) {
        const uint64_t p = m.off;
        if (m.type == 0x1) {                                     // dataspace, version 1 or 2
            const int ver = f.b[p];
            d.rank = f.b[p + 1];
            if (d.rank > 8) { err = "rank > 8"; return false; }
            const uint64_t q = p + (ver == 1 ? 8 : 4);
            for (int i = 0; i < d.rank; ++i) d.dims[i] = f.u(q + 8ull * i, 8);
            have_space = true;
        } else if (m.type == 0x3) {                              // datatype
            d.tclass = f.b[p] & 0x0f;
            const int bits0 = f.b[p + 1];
            d.esize = (int)f.u(p + 4, 4);
            if ((bits0 & 1) != 0) { err = "big-endian data"; return false; }
            if (d.tclass == 0) d.is_signed = (bits0 >> 3) & 1;
            else if (d.tclass != 1) { err = "only integer and floating-point datasets are arrays of the raw folder"; return false; }
            have_type = true;
        } else if (m.type == 0x8) {                              // data layout, version 3
            if (f.b[p] != 3) { err = "unsupported data layout message version"; return false; }
            d.layout = f.b[p + 1];
            if (d.layout == 1) { d.data_addr = f.u(p + 2, 8); d.data_size = f.u(p + 10, 8); }
            else if (d.layout == 2) {
                const int nd = f.b[p + 2];                       // rank + 1
                if (nd < 2 || nd > 9) { err = "bad chunk dimensionality"; return false; }
                d.chunk_nd = nd;
                d.data_addr = f.u(p + 3, 8);
                for (int i = 0; i < nd; ++i) {
                    d.chunk[i] = (uint32_t)f.u(p + 11 + 4ull * i, 4);
                    if (d.chunk[i] == 0) { err = "chunk dimension of size 0"; return false; }
                }
            } else { err = "compact datasets are not expected here"; return false; }
        } else if (m.type == 0xB) {                              // filter pipeline, version 1
            if (f.b[p] != 1) { err = "unsupported filter pipeline version"; return false; }
            const int nf = f.b[p + 1];
            uint64_t q = p + 8;
            for (int i = 0; i < nf; ++i) {
                const int id = (int)f.u(q, 2), nlen = (int)f.u(q + 2, 2), ncd = (int)f.u(q + 6, 2);
                if (id == 1) d.deflate = true;
                else if (id == 2) d.shuffle = true;
                else { err = "dataset uses a filter other than deflate / shuffle"; return false; }
                q += 8 + ((nlen + 7) / 8) * 8 + 4ull * ncd + ((ncd & 1) ? 4 : 0);
            }
        }
    }
    if (!have_space || !have_type || d.layout < 0) { err = "dataset header lacks dataspace / datatype / layout"; return false; }
    if (d.layout == 2 && d.chunk_nd != d.rank + 1) { err = "chunk dimensionality does not match the dataspace rank"; return false; }
    if (d.esize < 1 || d.esize > 16) { err = "element size outside 1..16 bytes"; return false; }
    return true;
}

void unshuffle(std::vector<uint8_t>& buf, int esize) {
    if (esize <= 1) return;
    const size_t n = buf.size() / esize;
    std::vector<uint8_t> out(buf.size());
    for (int b = 0; b < esize; ++b)
        for (size_t i = 0; i < n; ++i) out[i * esize + b] = buf[b * n + i];
    std::memcpy(out.data() + n * esize, buf.data() + n * esize, buf.size() - n * esize);
    buf.swap(out);
}

struct ChunkRef { uint64_t csize, mask, addr; uint64_t off[8]; };

// inflate + unshuffle one chunk and copy the part that lies inside the dataset, row by row of the fastest axis
bool place_chunk(const File& f, const Dataset& d, const ChunkRef& c, uint64_t chunk_bytes, std::vector<uint8_t>& buf, uint8_t* out, std::string& err) {
    const int R = d.rank;
    buf.resize(chunk_bytes);
    if (d.deflate && !(c.mask & 1)) {
        uLongf dl = (uLongf)chunk_bytes;
        if (uncompress(buf.data(), &dl, &f.b[c.addr], (uLong)c.csize) != Z_OK || dl != chunk_bytes) { err = "zlib: chunk does not inflate to the chunk size"; return false; }
    } else {
        if (c.csize != chunk_bytes) { err = "uncompressed chunk of the wrong size"; return false; }
        std::memcpy(buf.data(), &f.b[c.addr], chunk_bytes);
    }
    if (d.shuffle && !(c.mask & 2)) unshuffle(buf, d.esize);
    uint64_t idx[8] = {0};
    const uint64_t row = std::min<uint64_t>(d.chunk[R - 1], d.dims[R - 1] > c.off[R - 1] ? d.dims[R - 1] - c.off[R - 1] : 0);
    if (row == 0) return true;
    while (true) {
        bool inside = true;
        uint64_t src = 0, dst = 0;
        for (int k = 0; k < R - 1; ++k) {
            if (c.off[k] + idx[k] >= d.dims[k]) inside = false;
            src = src * d.chunk[k] + idx[k];
            dst = dst * d.dims[k] + c.off[k] + idx[k];
        }
        if (inside) std::memcpy(out + (dst * d.dims[R - 1] + c.off[R - 1]) * d.esize, buf.data() + src * d.chunk[R - 1] * d.esize, row * d.esize);
        int k = R - 2;
        while (k >= 0 && ++idx[k] == d.chunk[k]) idx[k--] = 0;
        if (k < 0) break;
    }
    return true;
}

// Chunks are independent (disjoint destination boxes): the B-tree walk collects them, then a few host threads inflate and place
// them in parallel -- a T = 12 tile is ~90 MB of deflate streams, 1 s on one core against 20 ms of GPU work per tile.
// TTC_IO_THREADS (default 8) bounds the threads of one call; job.iter_raw_tiles additionally reads several files at once.
bool read_chunks(const File& f, const Dataset& d, uint8_t* out, std::string& err) {
    const int R = d.rank;
    uint64_t chunk_elems = 1;
    for (int i = 0; i < R; ++i) chunk_elems *= d.chunk[i];
    const uint64_t chunk_bytes = chunk_elems * d.esize;
    // a corrupt chunk shape must not become a multi-gigabyte allocation in every inflate thread: HDF5 itself caps a chunk at 4 GiB - 1,
    // the job's chunks are < 1 MiB
    if (chunk_elems == 0 || chunk_bytes > (1ull << 31)) { err = "implausible chunk size (corrupt dataset layout?)"; return false; }
    std::vector<ChunkRef> refs;
    std::vector<uint64_t> stack{d.data_addr};
    while (!stack.empty()) {
        const uint64_t a = stack.back();
        stack.pop_back();
        if (!f.ok(a, 24) || std::memcmp(&f.b[a], "TREE", 4) != 0 || f.b[a + 4] != 1) { err = "bad chunk B-tree node"; return false; }
        const int level = f.b[a + 5], n = (int)f.u(a + 6, 2);
        const uint64_t ksz = 8 + 8ull * (R + 1);
        uint64_t p = a + 24;
        if (stack.size() + n > 1u << 22 || refs.size() > 1u << 24) { err = "chunk B-tree too large (cyclic?)"; return false; }
        for (int i = 0; i < n; ++i) {
            if (!f.ok(p, ksz + 8)) { err = "chunk B-tree entry past the end of the file"; return false; }
            const uint64_t csize = f.u(p, 4), mask = f.u(p + 4, 4), child = f.u(p + ksz, 8);
            if (level > 0) { stack.push_back(child); p += ksz + 8; continue; }
            ChunkRef c{csize, mask, child, {0}};
            for (int k = 0; k < R; ++k) c.off[k] = f.u(p + 8 + 8ull * k, 8);
            p += ksz + 8;
            if (!f.ok(child, csize)) { err = "chunk data past the end of the file"; return false; }
            refs.push_back(c);
        }
    }
    static const int max_threads = [] {
        const char* e = getenv("TTC_IO_THREADS");
        const int v = e ? atoi(e) : 8;
        return v < 1 ? 1 : (v > 64 ? 64 : v);
    }();
    const int nt = (int)std::min<size_t>((size_t)max_threads, std::max<size_t>(1, refs.size() / 4));
    if (nt <= 1) {
        std::vector<uint8_t> buf;
        for (const ChunkRef& c : refs) if (!place_chunk(f, d, c, chunk_bytes, buf, out, err)) return false;
        return true;
    }
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    std::mutex mu;
    auto work = [&] {
        std::string e;
        try {                                    // an exception that leaves a std::thread's function calls std::terminate
            std::vector<uint8_t> buf;
            for (size_t i = next.fetch_add(1); i < refs.size() && !failed.load(std::memory_order_relaxed); i = next.fetch_add(1))
                if (!place_chunk(f, d, refs[i], chunk_bytes, buf, out, e)) {
                    std::lock_guard<std::mutex> g(mu);
                    if (!failed.exchange(true)) err = e;
                    return;
                }
        } catch (const std::exception& ex) {
            std::lock_guard<std::mutex> g(mu);
            if (!failed.exchange(true)) err = std::string("inflate worker: ") + ex.what();
        }
    };
    std::vector<std::thread> pool;
    for (int i = 1; i < nt; ++i) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    return !failed.load();
}

// links of the group whose object header is at `oh` (is_group = false and no error when the object is not an old-style group)
bool group_links(const File& f, uint64_t oh, std::vector<std::pair<std::string, uint64_t>>& links, bool& is_group, std::string& err) {
    std::vector<Msg> msgs;
    is_group = false;
    if (!read_header(f, oh, msgs, err)) return false;
    for (const Msg& m : msgs)
        if (m.type == 0x11) {
            is_group = true;
            return list_group(f, f.u(m.off, 8), f.u(m.off + 8, 8), links, err);
        }
    return true;
}

// What hkl.load hands back first for the object at `oh`: the object itself when it is a dataset; for a group (the file root, or
// the container group hickle writes for a list / tuple / dict: `data` in hickle 4 / 5, `data_0` in hickle 3, children `data_i`)
// the member "data", else "data_0", else the first member that resolves, descended the same way.
bool default_dataset(const File& f, uint64_t oh, int depth, uint64_t& out, std::string& err) {
    if (depth > 8) { err = "groups nested deeper than 8 levels"; return false; }
    std::vector<std::pair<std::string, uint64_t>> links;
    bool is_group = false;
    if (!group_links(f, oh, links, is_group, err)) return false;
    if (!is_group) {
        Dataset t;
        if (!parse_dataset(f, oh, t, err)) return false;
        out = oh;
        return true;
    }
    for (const char* w : {"data", "data_0"})
        for (auto& l : links)
            if (l.first == w) return default_dataset(f, l.second, depth + 1, out, err);
    for (auto& l : links) {
        std::string e2;
        if (default_dataset(f, l.second, depth + 1, out, e2)) return true;
    }
    err = "group without a dataset";
    return false;
}

thread_local std::string g_hkl_err;     // per thread: loaders may read files concurrently

}  // namespace

extern "C" {

const char* ttc_read_hkl_error(void) { return g_hkl_err.c_str(); }

// name: '/'-separated path from the root ("data", "data/data_1"); NULL = what hkl.load returns first: "data", then "data_0", then
// the first dataset found, descending into container groups (hickle writes a list of arrays as a group of data_i datasets).
// h_out may be NULL (query shape / type only); cap_bytes = its capacity.
ttc_status ttc_read_hkl(const char* path, const char* name, void* h_out, size_t cap_bytes, int64_t* shape, int32_t* ndim,
                        int32_t* elem_size, int32_t* type_class, int32_t* is_signed) {
    auto fail = [](ttc_status s, const std::string& m) { g_hkl_err = m; return s; };
    if (!path || !shape || !ndim || !elem_size || !type_class || !is_signed) return fail(TTC_ERR_ARG, "read_hkl: null argument");
    File f;
    {
        const int fd = open(path, O_RDONLY);
        if (fd < 0) return fail(TTC_ERR_IO, std::string("read_hkl: cannot open ") + path);
        struct stat sb;
        if (fstat(fd, &sb) != 0 || sb.st_size < 0) { close(fd); return fail(TTC_ERR_IO, "read_hkl: cannot stat the file"); }
        // Default: map the file (no copy of ~90 MB per tile).  A mapped file that is TRUNCATED or rewritten while it is being parsed (a download
        // still in flight into temp/raw, some network filesystems) raises SIGBUS instead of an error return; TTC_HKL_READ=1 reads the bytes
        // into memory first (+ ~15 ms per tile), after which nothing another process does to the file can reach this one.
        static const bool use_read = [] { const char* e = getenv("TTC_HKL_READ"); return e && e[0] == '1'; }();
        if (sb.st_size > 0 && use_read) {
            try { f.owned.resize((size_t)sb.st_size); } catch (const std::exception&) { close(fd); return fail(TTC_ERR_NOMEM, "read_hkl: cannot allocate the file buffer"); }
            size_t got = 0;
            while (got < f.owned.size()) {
                const ssize_t r = pread(fd, f.owned.data() + got, f.owned.size() - got, (off_t)got);
                if (r <= 0) { close(fd); return fail(TTC_ERR_IO, "read_hkl: short read (file truncated while reading?)"); }
                got += (size_t)r;
            }
            f.b.p = f.owned.data(); f.b.n = f.owned.size();
        } else if (sb.st_size > 0) {
            void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { close(fd); return fail(TTC_ERR_IO, "read_hkl: cannot map the file"); }
            f.b.p = static_cast<const uint8_t*>(m); f.b.n = (size_t)sb.st_size;
        }
        close(fd);
    }
    static const uint8_t sig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
    if (f.b.size() < 96 || std::memcmp(f.b.data(), sig, 8) != 0) return fail(TTC_ERR_ARG, "read_hkl: not an HDF5 file");
    const int sver = f.b[8];
    if (sver > 1) return fail(TTC_ERR_ARG, "read_hkl: superblock version 2/3 (written with libver='latest') is not supported");
    if (f.b[13] != 8 || f.b[14] != 8) return fail(TTC_ERR_ARG, "read_hkl: only 8-byte offsets / lengths");
    const uint64_t root_entry = 24 + (sver == 1 ? 4 : 0) + 32;      // after base / free-space / EOF / driver addresses
    const uint64_t root_oh = f.u(root_entry + 8, 8);
    std::string err;
    uint64_t oh = kUndef;
    if (name && *name) {                                             // explicit path from the root: "data", "data/data_1", ...
        oh = root_oh;
        std::string rest(name);
        while (!rest.empty() && oh != kUndef) {
            const size_t cut = rest.find('/');
            const std::string part = rest.substr(0, cut);
            rest = cut == std::string::npos ? std::string() : rest.substr(cut + 1);
            if (part.empty()) continue;
            std::vector<std::pair<std::string, uint64_t>> links;
            bool is_group = false;
            if (!group_links(f, oh, links, is_group, err)) return fail(TTC_ERR_ARG, "read_hkl: " + err);
            uint64_t next = kUndef;
            if (is_group) for (auto& l : links) if (l.first == part) next = l.second;
            oh = next;
        }
        if (oh != kUndef && !default_dataset(f, oh, 0, oh, err)) oh = kUndef;   // a path that names a container: its first array
    } else {
        bool is_group = false;
        std::vector<std::pair<std::string, uint64_t>> links;
        if (!group_links(f, root_oh, links, is_group, err)) return fail(TTC_ERR_ARG, "read_hkl: root group: " + err);
        if (!is_group) return fail(TTC_ERR_ARG, "read_hkl: the root group has no symbol table (new-style group?)");
        if (!default_dataset(f, root_oh, 0, oh, err)) oh = kUndef;
    }
    Dataset d;
    if (oh == kUndef) return fail(TTC_ERR_ARG, std::string("read_hkl: dataset not found: ") + (name && *name ? name : "data / data_0") + (err.empty() ? "" : " (" + err + ")"));
    if (!parse_dataset(f, oh, d, err)) return fail(TTC_ERR_ARG, "read_hkl: " + err);
    *ndim = d.rank; *elem_size = d.esize; *type_class = d.tclass; *is_signed = d.is_signed;
    uint64_t total = (uint64_t)d.esize;
    for (int i = 0; i < d.rank; ++i) { shape[i] = (int64_t)d.dims[i]; total *= d.dims[i]; }
    if (!h_out) return TTC_OK;
    if (total > cap_bytes) return fail(TTC_ERR_ARG, "read_hkl: output buffer too small");
    if (d.rank == 0 || total == 0) { if (d.layout == 1 && d.data_addr != kUndef && f.ok(d.data_addr, total)) std::memcpy(h_out, &f.b[d.data_addr], total); return TTC_OK; }
    if (d.layout == 1) {
        if (d.data_addr == kUndef) { std::memset(h_out, 0, total); return TTC_OK; }
        if (!f.ok(d.data_addr, total)) return fail(TTC_ERR_ARG, "read_hkl: contiguous data past the end of the file");
        std::memcpy(h_out, &f.b[d.data_addr], total);
        return TTC_OK;
    }
    std::memset(h_out, 0, total);                                    // chunks that were never written read as the fill value 0
    if (d.data_addr == kUndef) return TTC_OK;
    if (!read_chunks(f, d, static_cast<uint8_t*>(h_out), err)) return fail(TTC_ERR_ARG, "read_hkl: " + err);
    return TTC_OK;
}

}  // extern "C"
