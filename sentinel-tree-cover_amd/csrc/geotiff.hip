// GeoTIFF writer for the finished uint8 raster: the file write_tif produces through rasterio
// (src/downloading/io.py:229-263: driver GTiff, one uint8 band, compress = lzw, CRS "+proj=longlat +datum=WGS84",
// transform from_bounds(west, south, east, north, width, height)).  Host code: the raster
// leaves the GPU as 0.38 MB of uint8; TIFF LZW is a serial byte-stream code, it stays on the host.
//
// Layout: classic little-endian TIFF, one IFD, strips of 8 rows (a multiple of GDAL's default), LZW with the TIFF
// conventions (MSB-first codes, ClearCode 256 first, EOI 257, "early change": the code width grows one code early),
// no predictor (GDAL's default for compress=lzw).  Georeferencing tags: ModelPixelScale (33550), ModelTiepoint (33922),
// GeoKeyDirectory (34735) = {GTModelType geographic, GTRasterType PixelIsArea, GeographicType EPSG:4326}.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ttc_internal.h"

namespace {

struct BitWriter {
    std::vector<uint8_t>& out;
    uint32_t acc = 0;
    int nbits = 0;
    explicit BitWriter(std::vector<uint8_t>& o) : out(o) {}
    void put(uint32_t code, int width) {
        acc = (acc << width) | code;
        nbits += width;
        while (nbits >= 8) { out.push_back((uint8_t)(acc >> (nbits - 8))); nbits -= 8; }
        acc &= (1u << nbits) - 1u;
    }
    void flush() { if (nbits > 0) { out.push_back((uint8_t)(acc << (8 - nbits))); nbits = 0; acc = 0; } }
};

// TIFF 6.0 section 13.  Dictionary: hash of (prefix code, byte) -> code.
void lzw_encode(const uint8_t* src, size_t n, std::vector<uint8_t>& out) {
    constexpr int kHash = 1 << 13;
    std::vector<int32_t> key(kHash), val(kHash);
    auto reset = [&] { std::fill(key.begin(), key.end(), -1); };
    BitWriter bw(out);
    int next = 258, width = 9;
    reset();
    bw.put(256, width);
    if (n == 0) { bw.put(257, width); bw.flush(); return; }
    int prefix = src[0];
    for (size_t i = 1; i < n; ++i) {
        const int c = src[i];
        const int32_t k = (prefix << 8) | c;
        int h = (int)(((uint32_t)k * 2654435761u) >> 19) & (kHash - 1);
        int found = -1;
        while (key[h] != -1) {
            if (key[h] == k) { found = val[h]; break; }
            h = (h + 1) & (kHash - 1);
        }
        if (found >= 0) { prefix = found; continue; }
        bw.put((uint32_t)prefix, width);
        key[h] = k; val[h] = next++;
        if (next == 512 || next == 1024 || next == 2048) ++width;       // the decoder's table lags by one entry: "early change"
        if (next == 4094) {                                               // table full: clear
            bw.put(256, width);
            reset(); next = 258; width = 9;
        }
        prefix = c;
    }
    bw.put((uint32_t)prefix, width);
    ++next;                                                               // the decoder adds an entry for this code too
    if (next == 512 || next == 1024 || next == 2048) ++width;
    if (next == 4094) { bw.put(256, width); width = 9; }
    bw.put(257, width);
    bw.flush();
}

struct Entry { uint16_t tag, type; uint32_t count, value; };

template <typename T>
void append(std::vector<uint8_t>& b, const T& v) {
    const uint8_t* p = reinterpret_cast<const uint8_t*>(&v);
    b.insert(b.end(), p, p + sizeof(T));
}

}  // namespace

ttc_status ttc_write_geotiff_u8(const char* path, const uint8_t* h_raster, int32_t rows, int32_t cols, double west, double south,
                                double east, double north) {
    if (!path || !h_raster || rows < 1 || cols < 1 || !(east > west) || !(north > south)) return TTC_ERR_ARG;
    const int rps = 8;
    const int nstrips = (rows + rps - 1) / rps;
    std::vector<uint8_t> body;                      // everything after the 8-byte header, offsets are body-relative + 8
    std::vector<uint32_t> offs(nstrips), cnts(nstrips);
    for (int s = 0; s < nstrips; ++s) {
        const int r0 = s * rps, nr = std::min(rps, rows - r0);
        offs[s] = (uint32_t)body.size() + 8;
        lzw_encode(h_raster + (size_t)r0 * cols, (size_t)nr * cols, body);
        cnts[s] = (uint32_t)body.size() + 8 - offs[s];
        if (body.size() & 1) body.push_back(0);
    }
    auto blob = [&](const void* p, size_t n) { const uint32_t o = (uint32_t)body.size() + 8; body.insert(body.end(), (const uint8_t*)p, (const uint8_t*)p + n); if (body.size() & 1) body.push_back(0); return o; };
    const uint32_t off_offs = nstrips > 1 ? blob(offs.data(), 4 * offs.size()) : offs[0];
    const uint32_t off_cnts = nstrips > 1 ? blob(cnts.data(), 4 * cnts.size()) : cnts[0];
    const double scale[3] = {(east - west) / cols, (north - south) / rows, 0.0};        // rasterio.transform.from_bounds
    const double tie[6] = {0, 0, 0, west, north, 0};
    const uint16_t keys[16] = {1, 1, 0, 3, 1024, 0, 1, 2, 1025, 0, 1, 1, 2048, 0, 1, 4326};
    const uint32_t off_scale = blob(scale, sizeof scale), off_tie = blob(tie, sizeof tie), off_keys = blob(keys, sizeof keys);
    const Entry ifd[] = {
        {256, 3, 1, (uint32_t)cols}, {257, 3, 1, (uint32_t)rows}, {258, 3, 1, 8}, {259, 3, 1, 5 /* LZW */}, {262, 3, 1, 1 /* BlackIsZero */},
        {273, 4, (uint32_t)nstrips, off_offs}, {277, 3, 1, 1}, {278, 3, 1, (uint32_t)rps}, {279, 4, (uint32_t)nstrips, off_cnts},
        {284, 3, 1, 1}, {339, 3, 1, 1 /* unsigned */}, {33550, 12, 3, off_scale}, {33922, 12, 6, off_tie}, {34735, 3, 16, off_keys}};
    const uint32_t ifd_off = (uint32_t)body.size() + 8;
    append(body, (uint16_t)(sizeof ifd / sizeof ifd[0]));
    for (const Entry& e : ifd) { append(body, e.tag); append(body, e.type); append(body, e.count); append(body, e.value); }
    append(body, (uint32_t)0);
    FILE* f = std::fopen(path, "wb");
    if (!f) return TTC_ERR_IO;
    const uint8_t hdr[4] = {'I', 'I', 42, 0};
    bool ok = std::fwrite(hdr, 1, 4, f) == 4 && std::fwrite(&ifd_off, 4, 1, f) == 1 &&
              std::fwrite(body.data(), 1, body.size(), f) == body.size();
    ok = (std::fclose(f) == 0) && ok;
    return ok ? TTC_OK : TTC_ERR_IO;
}
