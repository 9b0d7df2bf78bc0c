"""Pins oracle/restate_model.py::DSen2Lite against the reference's frozen graph ITSELF.

TensorFlow is absent, but models-release/supres-40k-swir/superresolve_graph.pb is data: this script decodes the GraphDef
(tools/extract_dsen2.py's protobuf wire reader) and INTERPRETS its 61 nodes in stored (topological) order with torch --
Placeholder, Const, Identity, MirrorPad, Conv2D, BiasAdd, Relu, Mul, Add, Tanh with the attributes the nodes carry
(padding VALID, NHWC, mode REFLECT, the int32 paddings constants) -- so the topology, the constants and the weights all
come from the reference file, none from the restatement.  Output tensor "Add_2" (job.py:1794) for seeded inputs ->
tests/golden/dsen2_graph.npz.

Build container only:   python tools/gen_golden_dsen2.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import extract_dsen2 as X  # noqa: E402

PB = "/root/reference/models-release/supres-40k-swir/superresolve_graph.pb"


def const_any(node_bytes):
    """TensorProto of a Const node's `value` attr, float32 or int32"""
    for f2, _, v2 in X.fields(node_bytes):
        if f2 != 5:
            continue
        key = val = None
        for f3, _, v3 in X.fields(v2):
            if f3 == 1:
                key = v3.decode()
            elif f3 == 2:
                val = v3
        if key != "value":
            continue
        for f4, _, v4 in X.fields(val):
            if f4 != 8:
                continue
            dtype, shape, content, ivals = None, [], b"", []
            for f, wt, v in X.fields(v4):
                if f == 1:
                    dtype = v
                elif f == 2:
                    for g2, _, w2 in X.fields(v):
                        if g2 == 2:
                            size = 0
                            for g3, _, w3 in X.fields(w2):
                                if g3 == 1:
                                    size = w3
                            shape.append(size)
                elif f == 4:
                    content = v
                elif f == 7:
                    ivals.append(v)
            if dtype == 3:
                arr = np.frombuffer(content, dtype="<i4").copy() if content else np.array(ivals, dtype=np.int32)
                return arr.reshape(shape)
    return None


def run_graph(feeds, fetch="Add_2"):
    data = open(PB, "rb").read()
    raw_nodes = [v for f, _, v in X.fields(data) if f == 1]
    nodes = X.parse_graph(PB)
    assert len(nodes) == len(raw_nodes)
    val = {}
    for node, raw in zip(nodes, raw_nodes):
        op, name, ins = node["op"], node["name"], [val[i.split(":")[0]] for i in node["input"]]
        if op == "Placeholder":
            out = torch.as_tensor(feeds[name], dtype=torch.float64)
        elif op == "Const":
            t = node["tensor"]
            out = torch.as_tensor(t, dtype=torch.float64) if t is not None else torch.as_tensor(const_any(raw))
        elif op == "Identity":
            out = ins[0]
        elif op == "MirrorPad":
            assert node["attr"]["mode"] == "REFLECT"
            p = ins[1].tolist()
            assert p[0] == [0, 0] and p[3] == [0, 0]
            out = torch.nn.functional.pad(ins[0].permute(0, 3, 1, 2), (p[2][0], p[2][1], p[1][0], p[1][1]), mode="reflect").permute(0, 2, 3, 1)
        elif op == "Conv2D":
            assert node["attr"]["padding"] == "VALID" and node["attr"]["data_format"] == "NHWC"
            out = torch.nn.functional.conv2d(ins[0].permute(0, 3, 1, 2), ins[1].permute(3, 2, 0, 1)).permute(0, 2, 3, 1)
        elif op == "BiasAdd":
            out = ins[0] + ins[1]
        elif op == "Relu":
            out = torch.relu(ins[0])
        elif op == "Mul":
            out = ins[0] * ins[1]
        elif op == "Add":
            out = ins[0] + ins[1]
        elif op == "Tanh":
            out = torch.tanh(ins[0])
        else:
            raise NotImplementedError(op)
        val[name] = out
    return val[fetch].numpy(), len(nodes)


def main():
    rng = np.random.default_rng(42)
    out = {}
    for tag, shape in (("a", (2, 24, 24, 10)), ("b", (1, 118, 118, 10)), ("c", (3, 7, 5, 10))):
        x = rng.random(shape).astype(np.float32)
        bil = rng.random(shape[:3] + (6,)).astype(np.float32)
        y, n = run_graph({"Placeholder": x, "Placeholder_1": bil})
        out[tag + "_x"], out[tag + "_bil"], out[tag + "_y"] = x, bil, y
        print(tag, shape, "->", y.shape, "nodes", n)
    path = os.path.join(ROOT, "tests", "golden", "dsen2_graph.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
