#!/usr/bin/env python3
"""Fixture extractor (TEST INFRASTRUCTURE - runs only in the build container).

Reads the reference's *data* files under /root/reference (never its sources) and writes
small numeric fixtures under tests/golden/:

  g45_init.npz      initial weights + baked constants pulled out of the TF MetaGraphDef text proto
                    resources/ppo_cl/graphs/ppo_cpp_[4_5]_lr_0.0004_cr_0.1610_ent_0.0007.meta.txt  ("G")
  ckpt71.npz        the 15 tensors of the trained [4,5] checkpoint
                    resources/ppo_cl/2019-08-20_21_13_01_2859_0.pkl.71.data-00000-of-00001 (+ .index)
  ckpt71_stats.json hyper-parameters and obs/ret running statistics from ...pkl.71.json

Fixtures are data (numbers), not reference source text.  /root/reference does not exist on the
GPU box; nothing at test/bench time reads it - only these committed fixtures.

Usage:  python oracle/extract_fixtures.py [--ref /root/reference]
"""
import argparse
import ast
import glob
import json
import os
import re
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")


# --------------------------------------------------------------------------------------------
# Minimal brace-matching reader for protobuf *text* format (no schema needed).
# --------------------------------------------------------------------------------------------
def parse_block(lines, i=0):
    """Return (list of (key, value|sublist), next_index)."""
    out = []
    n = len(lines)
    while i < n:
        l = lines[i].strip()
        if l == "}":
            return out, i + 1
        if l.endswith("{"):
            sub, i = parse_block(lines, i + 1)
            out.append((l[:-1].strip(), sub))
        elif l:
            k, v = l.split(":", 1)
            out.append((k.strip(), v.strip()))
            i += 1
        else:
            i += 1
    return out, i


def first(block, key, default=None):
    for k, v in block:
        if k == key:
            return v
    return default


def every(block, key):
    return [v for k, v in block if k == key]


def unquote(s):
    return ast.literal_eval(s)


def tensor_from_attr(node):
    """Decode the Const node's `value` attr into a numpy array (float32 / int32)."""
    for attr in every(node, "attr"):
        if unquote(first(attr, "key")) != "value":
            continue
        t = first(first(attr, "value"), "tensor")
        dtype = first(t, "dtype")
        shape_blk = first(t, "tensor_shape", [])
        dims = [int(first(d, "size", "0")) for d in every(shape_blk, "dim")]
        np_dt = {"DT_FLOAT": "<f4", "DT_INT32": "<i4"}.get(dtype)
        if np_dt is None:
            return None
        content = first(t, "tensor_content")
        if content is not None:
            raw = ast.literal_eval("b" + content)
            arr = np.frombuffer(raw, dtype=np_dt).copy()
        else:
            key = "float_val" if dtype == "DT_FLOAT" else "int_val"
            vals = [float(v) if dtype == "DT_FLOAT" else int(v) for v in every(t, key)]
            arr = np.array(vals, dtype=np_dt)
            count = int(np.prod(dims)) if dims else 1
            if arr.size == 1 and count > 1:
                arr = np.full(count, arr[0], dtype=np_dt)
        return arr.reshape(dims) if dims else arr.reshape(())
    return None


def load_graph_nodes(path):
    with open(path, "r") as f:
        lines = f.read().split("\n")
    top, _ = parse_block(lines)
    gd = first(top, "graph_def")
    nodes = {}
    for n in every(gd, "node"):
        nodes[unquote(first(n, "name"))] = n
    return nodes


# --------------------------------------------------------------------------------------------
# TF bundle (.index is a LevelDB-style table; entries are BundleEntryProto)
# --------------------------------------------------------------------------------------------
def read_varint(buf, pos):
    result, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def parse_bundle_entry(buf):
    """BundleEntryProto: 1 dtype, 2 shape{2 dim{1 size}}, 3 shard_id, 4 offset, 5 size, 6 crc32c."""
    pos, out = 0, {"shape": [], "offset": 0, "size": 0}
    while pos < len(buf):
        tag, pos = read_varint(buf, pos)
        fno, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = read_varint(buf, pos)
            if fno == 1:
                out["dtype"] = v
            elif fno == 4:
                out["offset"] = v
            elif fno == 5:
                out["size"] = v
        elif wt == 5:
            out["crc32c"] = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        elif wt == 2:
            ln, pos = read_varint(buf, pos)
            sub = buf[pos:pos + ln]
            pos += ln
            if fno == 2:                                  # TensorShapeProto
                sp = 0
                while sp < len(sub):
                    t2, sp = read_varint(sub, sp)
                    if (t2 >> 3) == 2 and (t2 & 7) == 2:  # dim
                        l2, sp = read_varint(sub, sp)
                        dim = sub[sp:sp + l2]
                        sp += l2
                        dp = 0
                        while dp < len(dim):
                            t3, dp = read_varint(dim, dp)
                            if (t3 & 7) == 0:
                                v3, dp = read_varint(dim, dp)
                                if (t3 >> 3) == 1:
                                    out["shape"].append(v3)
                            else:
                                l3, dp = read_varint(dim, dp)
                                dp += l3
                    else:
                        if (t2 & 7) == 0:
                            _, sp = read_varint(sub, sp)
                        else:
                            l2, sp = read_varint(sub, sp)
                            sp += l2
        else:
            raise ValueError("unexpected wire type %d" % wt)
    return out


def read_bundle_index(path):
    """Walk the single data block of the SSTable: prefix-compressed (shared, non_shared, vlen) records."""
    buf = open(path, "rb").read()
    # footer = last 48 bytes: metaindex handle, index handle (varints), padding, 8-byte magic
    footer = buf[-48:]
    p = 0
    _, p = read_varint(footer, p)
    _, p = read_varint(footer, p)
    idx_off, p = read_varint(footer, p)
    idx_size, p = read_varint(footer, p)
    # index block: one entry per data block -> handle(offset,size)
    blocks = []
    ib = buf[idx_off:idx_off + idx_size]
    n_restarts = struct.unpack_from("<I", ib, len(ib) - 4)[0]
    end = len(ib) - 4 - 4 * n_restarts
    p, key = 0, b""
    while p < end:
        shared, p = read_varint(ib, p)
        non_shared, p = read_varint(ib, p)
        vlen, p = read_varint(ib, p)
        key = key[:shared] + ib[p:p + non_shared]
        p += non_shared
        val = ib[p:p + vlen]
        p += vlen
        o, q = read_varint(val, 0)
        s, q = read_varint(val, q)
        blocks.append((o, s))
    entries = {}
    for (o, s) in blocks:
        db = buf[o:o + s]
        n_restarts = struct.unpack_from("<I", db, len(db) - 4)[0]
        end = len(db) - 4 - 4 * n_restarts
        p, key = 0, b""
        while p < end:
            shared, p = read_varint(db, p)
            non_shared, p = read_varint(db, p)
            vlen, p = read_varint(db, p)
            key = key[:shared] + db[p:p + non_shared]
            p += non_shared
            val = db[p:p + vlen]
            p += vlen
            if key == b"":
                continue                                   # header entry
            entries[key.decode()] = parse_bundle_entry(val)
    return entries


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    os.makedirs(GOLDEN, exist_ok=True)

    gpath = glob.glob(os.path.join(args.ref, "resources/ppo_cl/graphs/*.meta.txt"))[0]
    nodes = load_graph_nodes(gpath)
    print("graph nodes:", len(nodes))

    out = {}
    var_names = ["pi_fc0/w", "pi_fc0/b", "vf_fc0/w", "vf_fc0/b", "pi_fc1/w", "pi_fc1/b", "vf_fc1/w", "vf_fc1/b",
                 "vf/w", "vf/b", "pi/w", "pi/b", "pi/logstd", "q/w", "q/b"]
    for v in var_names:
        cands = [n for n in nodes if n.startswith("model/" + v + "/Initializer/")
                 and first(nodes[n], "op") and unquote(first(nodes[n], "op")) == "Const"]
        arr = None
        for c in cands:
            a = tensor_from_attr(nodes[c])
            if a is not None and a.dtype == np.float32 and (arr is None or a.size > arr.size):
                arr = a
        if arr is None:
            raise SystemExit("no initializer const for " + v + " : " + str(cands))
        # zeros initializers are stored as scalar + shape; recover the shape from the variable node
        vnode = nodes["model/" + v]
        for attr in every(vnode, "attr"):
            if unquote(first(attr, "key")) == "shape":
                dims = [int(first(d, "size")) for d in every(first(first(attr, "value"), "shape"), "dim")]
                if arr.size == 1 and int(np.prod(dims)) > 1:
                    arr = np.full(dims, float(arr.reshape(-1)[0]), dtype=np.float32)
                arr = arr.reshape(dims)
        out[v] = arr.astype(np.float32)
        print("  init %-10s %-10s |x|max=%.4f" % (v, arr.shape, float(np.abs(arr).max())))

    # baked scalar constants (survey section 5 'config / flags' cites their line numbers)
    def scalar_of(name):
        a = tensor_from_attr(nodes[name])
        return float(a.reshape(-1)[0])

    consts = {}
    for name in nodes:
        n = nodes[name]
        if unquote(first(n, "op")) != "Const":
            continue
        a = tensor_from_attr(n)
        if a is None or a.size != 1 or a.dtype != np.float32:
            continue
        consts[name] = float(a.reshape(-1)[0])
    # keep the ones the formula sheet needs (survey section 5 'config / flags' row, App. B)
    pick = {}
    wanted = ["loss/mul_4/y", "loss/mul_5/y",                       # ent_coef, vf_coef
              "ppo2/_train/beta1", "ppo2/_train/beta2", "ppo2/_train/epsilon",
              "beta1_power/initial_value", "beta2_power/initial_value",
              "loss/clip_by_global_norm/mul/x", "loss/clip_by_global_norm/truediv/x",
              "loss/clip_by_global_norm/truediv_1/y", "loss/global_norm/Const_1"]
    for name in wanted:
        pick[name] = consts[name]
    # order of the 13 ApplyAdam ops = flat parameter order (survey App. B)
    adam_order = []
    with open(gpath) as f:
        for line in f:
            m = re.search(r'name: "ppo2/_train/update_model/(.*)/ApplyAdam"', line)
            if m:
                adam_order.append(m.group(1))
    print("  ApplyAdam order:", adam_order)
    out["adam_order"] = np.array(adam_order, dtype=object)
    # global-norm stack order (inputs of loss/global_norm/stack)
    gn = nodes["loss/global_norm/stack"]
    gn_inputs = [unquote(v) for v in every(gn, "input")]
    print("  global_norm stack:", gn_inputs)
    for k in sorted(pick):
        print("  const %-60s %r" % (k, pick[k]))
    out["const_names"] = np.array(sorted(pick), dtype=object)
    out["const_values"] = np.array([pick[k] for k in sorted(pick)], dtype=np.float64)
    # op histogram (sanity, cited by survey App. D)
    hist = {}
    for n in nodes.values():
        op = unquote(first(n, "op"))
        hist[op] = hist.get(op, 0) + 1
    print("  ops: MatMul=%d ApplyAdam=%d Tanh=%d TanhGrad=%d" % (hist.get("MatMul", 0), hist.get("ApplyAdam", 0),
                                                               hist.get("Tanh", 0), hist.get("TanhGrad", 0)))
    np.savez(os.path.join(GOLDEN, "g45_init.npz"), **{k.replace("/", "__"): v for k, v in out.items()},
             allow_pickle=True)

    # ---- checkpoint -------------------------------------------------------------------------
    idx = glob.glob(os.path.join(args.ref, "resources/ppo_cl/*.index"))[0]
    dat = glob.glob(os.path.join(args.ref, "resources/ppo_cl/*.data-00000-of-00001"))[0]
    entries = read_bundle_index(idx)
    raw = open(dat, "rb").read()
    ck = {}
    for name in sorted(entries):
        e = entries[name]
        arr = np.frombuffer(raw[e["offset"]:e["offset"] + e["size"]], dtype="<f4").copy().reshape(e["shape"])
        short = name.replace("model/", "")
        ck[short.replace("/", "__")] = arr
        print("  ckpt %-10s off=%4d size=%4d shape=%s" % (short, e["offset"], e["size"], e["shape"]))
    np.savez(os.path.join(GOLDEN, "ckpt71.npz"), **ck)

    # the checkpoint itself is a DATA file of the reference tree: kept as a fixture so that the product's bundle
    # reader / writer can be tested byte for byte where /root/reference does not exist
    import shutil
    shutil.copyfile(idx, os.path.join(GOLDEN, "ckpt71.index"))
    shutil.copyfile(dat, os.path.join(GOLDEN, "ckpt71.data-00000-of-00001"))
    js = glob.glob(os.path.join(args.ref, "resources/ppo_cl/*.json"))[0]
    stats = json.load(open(js))
    json.dump(stats, open(os.path.join(GOLDEN, "ckpt71_stats.json"), "w"), indent=1, sort_keys=True)
    print("wrote fixtures to", GOLDEN)


if __name__ == "__main__":
    sys.setrecursionlimit(10000)
    main()
