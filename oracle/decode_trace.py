"""Decode the `add_graph` trace embedded in the reference's TensorBoard event files.

TEST INFRASTRUCTURE (oracle side).  Runs only in the build container, where
/root/reference exists.  It reads one of the reference's committed fixtures
(`test_logs/*/events.out.tfevents.*`, written by
`src/visualization/tensorboard_logger.py:79-83` <- `src/test_system.py:190`) and
extracts *derived data*: the ordered list of aten ops with their module scope,
output shapes and conv/BN/pool hyper-parameters.  The result is committed as
`tests/golden/unet_r50_trace.json` and pins `oracle.unet_ref.UnetRef('resnet50')`.

No tensorboard / tensorflow needed: a ~100-line protobuf wire-format reader.
"""
import json
import struct
import sys


def _varint(b, i):
    r = 0
    s = 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        if not c & 0x80:
            return r, i
        s += 7


def _fields(b):
    """Yield (field_number, wire_type, value) for one protobuf message."""
    i = 0
    n = len(b)
    while i < n:
        key, i = _varint(b, i)
        f, w = key >> 3, key & 7
        if w == 0:
            v, i = _varint(b, i)
        elif w == 1:
            v = b[i:i + 8]
            i += 8
        elif w == 2:
            ln, i = _varint(b, i)
            v = b[i:i + ln]
            i += ln
        elif w == 5:
            v = b[i:i + 4]
            i += 4
        else:
            raise ValueError("wire type %d" % w)
        yield f, w, v


def _records(path):
    with open(path, "rb") as fh:
        data = fh.read()
    i = 0
    while i + 12 <= len(data):
        (ln,) = struct.unpack("<Q", data[i:i + 8])
        i += 12
        yield data[i:i + ln]
        i += ln + 4


def _shape_list(attr_value):
    """AttrValue.list(1) -> repeated shape(7) -> repeated dim(2) -> size(1)."""
    shapes = []
    for f, w, v in _fields(attr_value):
        if f == 1 and w == 2:  # list
            for f2, w2, v2 in _fields(v):
                if f2 == 7 and w2 == 2:  # TensorShapeProto
                    dims = []
                    for f3, w3, v3 in _fields(v2):
                        if f3 == 2 and w3 == 2:
                            for f4, w4, v4 in _fields(v3):
                                if f4 == 1 and w4 == 0:
                                    dims.append(v4)
                    shapes.append(dims)
    return shapes


def _node(b):
    name = op = None
    inputs = []
    attrs = {}
    for f, w, v in _fields(b):
        if f == 1:
            name = v.decode()
        elif f == 2:
            op = v.decode()
        elif f == 3:
            inputs.append(v.decode())
        elif f == 5:
            k = val = None
            for f2, w2, v2 in _fields(v):
                if f2 == 1:
                    k = v2.decode()
                elif f2 == 2:
                    val = v2
            attrs[k] = val
    out = {"name": name, "op": op, "inputs": inputs}
    if "_output_shapes" in attrs:
        out["shapes"] = _shape_list(attrs["_output_shapes"])
    if "attr" in attrs:
        for f, w, v in _fields(attrs["attr"]):
            if f == 2:
                out["attr"] = v.decode()
    return out


def graph_nodes(path):
    for rec in _records(path):
        for f, w, v in _fields(rec):
            if f == 4 and w == 2:  # Event.graph_def
                return [_node(nv) for nf, nw, nv in _fields(v) if nf == 1 and nw == 2]
    return None


def _const_value(node):
    a = node.get("attr", "")
    # strings look like "{ value : 2}"
    if "value" in a:
        try:
            return json.loads(a.split(":", 1)[1].strip(" }"))
        except Exception:
            return a
    return None


def summarise(nodes):
    by = {n["name"]: n for n in nodes}

    def lst(name):
        n = by.get(name)
        if n is None:
            return None
        if n["op"] == "prim::ListConstruct":
            return [_const_value(by[i]) for i in n["inputs"]]
        if n["op"] == "prim::Constant":
            return _const_value(n)
        return None

    ops = []
    for n in nodes:
        op = n["op"]
        if not op.startswith("aten::"):
            continue
        if op in ("aten::_convolution", "aten::batch_norm", "aten::relu_", "aten::relu", "aten::add_", "aten::add",
                  "aten::max_pool2d", "aten::upsample_nearest2d", "aten::upsample_bilinear2d", "aten::cat"):
            scope = n["name"].rsplit("/", 1)[0]
            e = {"op": op[6:], "scope": scope, "out": (n.get("shapes") or [None])[0]}
            ins = n["inputs"]
            if op == "aten::_convolution":
                x = by.get(ins[0])
                e["in"] = (x.get("shapes") or [None])[0] if x else None
                b = by.get(ins[2])
                e["bias"] = bool(b and b["op"] != "prim::Constant")
                e["stride"], e["padding"], e["dilation"] = lst(ins[3]), lst(ins[4]), lst(ins[5])
                e["transposed"], e["groups"] = lst(ins[6]), lst(ins[8])
            elif op == "aten::batch_norm":
                e["training"], e["momentum"], e["eps"] = lst(ins[5]), lst(ins[6]), lst(ins[7])
            elif op == "aten::max_pool2d":
                e["kernel"], e["stride"], e["padding"], e["dilation"], e["ceil_mode"] = (lst(i) for i in ins[1:6])
            elif op.startswith("aten::upsample"):
                e["scales"] = [lst(i) for i in ins[1:]]
            elif op == "aten::cat":
                src = by.get(ins[0])
                e["n_inputs"] = len(src["inputs"]) if src else None
                e["in_shapes"] = [(by[i].get("shapes") or [None])[0] for i in src["inputs"]] if src else None
                e["dim"] = lst(ins[1])
            ops.append(e)
    return ops


if __name__ == "__main__":
    src = sys.argv[1]
    dst = sys.argv[2]
    nodes = graph_nodes(src)
    ops = summarise(nodes)
    with open(dst, "w") as fh:
        json.dump({"source": src.replace("/root/reference/", ""), "ops": ops}, fh, indent=0)
    from collections import Counter
    print(Counter(o["op"] for o in ops))
