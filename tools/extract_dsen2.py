"""Extract the public DSen2-lite weights from the reference's frozen GraphDef
(models-release/supres-40k-swir/superresolve_graph.pb) WITHOUT TensorFlow, by
decoding the protobuf wire format directly.

Runs only where /root/reference exists.  Output: an .npz of
{in_conv,01_conv,02_conv,11_conv,12_conv,out_conv}/{kernel,bias} in TF layout
(HWIO).  `--dump` prints the node list (name, op, inputs) used to restate the
graph in oracle/restate_model.py::DSen2Lite.
"""
import struct
import sys

import numpy as np


def _varint(b, i):
    r = s = 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        s += 7
        if not c & 0x80:
            return r, i


def fields(b):
    """Yield (field_no, wire_type, value) for one message."""
    i = 0
    while i < len(b):
        key, i = _varint(b, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(b, i)
        elif wt == 1:
            v = b[i:i + 8]
            i += 8
        elif wt == 2:
            n, i = _varint(b, i)
            v = b[i:i + n]
            i += n
        elif wt == 5:
            v = b[i:i + 4]
            i += 4
        else:
            raise ValueError(f"wire type {wt}")
        yield f, wt, v


def parse_tensor(b):
    dtype, shape, content, fvals = None, [], b"", []
    for f, wt, v in fields(b):
        if f == 1:
            dtype = v
        elif f == 2:
            for f2, _, v2 in fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in fields(v2):
                        if f3 == 1:
                            size = v3
                    shape.append(size)
        elif f == 4:
            content = v
        elif f == 5:
            if wt == 2:
                fvals += list(struct.unpack(f"<{len(v) // 4}f", v))
            else:
                fvals.append(struct.unpack("<f", v)[0])
    if dtype != 1:      # DT_FLOAT
        return None
    if content:
        arr = np.frombuffer(content, dtype="<f4").copy()
    else:
        arr = np.array(fvals, dtype=np.float32)
    n = int(np.prod(shape)) if shape else 1
    if arr.size == 1 and n > 1:
        arr = np.full(n, arr[0], dtype=np.float32)
    return arr.reshape(shape)


def parse_graph(path):
    nodes = []
    data = open(path, "rb").read()
    for f, wt, v in fields(data):
        if f != 1:
            continue
        node = {"name": "", "op": "", "input": [], "tensor": None, "attr": {}}
        for f2, wt2, v2 in fields(v):
            if f2 == 1:
                node["name"] = v2.decode()
            elif f2 == 2:
                node["op"] = v2.decode()
            elif f2 == 3:
                node["input"].append(v2.decode())
            elif f2 == 5:
                key, val = None, None
                for f3, _, v3 in fields(v2):
                    if f3 == 1:
                        key = v3.decode()
                    elif f3 == 2:
                        val = v3
                if key == "value" and val is not None:
                    for f4, _, v4 in fields(val):
                        if f4 == 8:
                            node["tensor"] = parse_tensor(v4)
                elif key is not None and val is not None:
                    for f4, wt4, v4 in fields(val):
                        if f4 == 2:
                            node["attr"][key] = v4.decode(errors="replace")
        nodes.append(node)
    return nodes


def main():
    src = "/root/reference/models-release/supres-40k-swir/superresolve_graph.pb"
    out = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else None
    nodes = parse_graph(src)
    if "--dump" in sys.argv:
        for n in nodes:
            t = "" if n["tensor"] is None else f" const{list(n['tensor'].shape)}"
            print(f"{n['op']:12s} {n['name']:44s} <- {n['input']}{t} {n['attr']}")
    weights = {}
    for n in nodes:
        if n["op"] == "Const" and n["tensor"] is not None and n["tensor"].ndim in (1, 4):
            parts = n["name"].split("/")
            if parts[-1] in ("kernel", "bias"):
                weights[f"{parts[0]}/{parts[-1]}"] = n["tensor"]
    print({k: v.shape for k, v in weights.items()}, sum(v.size for v in weights.values()))
    if out:
        np.savez(out, **weights)


if __name__ == "__main__":
    main()
