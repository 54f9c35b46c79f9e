"""A second witness for CH05 / CH06 from the reference's own BINARIES.  Build container only: reads
/root/reference/RTCHAP06/Shaders/raytrace06.comp.spv and RTCHAP05/RTCHAP05/Shaders/raytrace05.comp.spv -- the compiled shaders
RTCHAP06/main.cpp:154 and RTCHAP05/RTCHAP05/main.cpp:136 actually load -- and writes tests/golden/spirv_witness.json with DERIVED
facts only (no SPIR-V words, no disassembly text):

  * per function: the ordered list of arithmetic opcode names, the GLSL.std.450 extended instructions used, whether any result id
    carries the NoContraction decoration; the module's float constants; LocalSize;
  * the VALUE each function returns / main stores, as an expression tree over the shader's inputs (gl_GlobalInvocationID, the UBO's
    five members), obtained by executing the SPIR-V symbolically (loads / stores / access chains / calls / structured branches
    folded away): operation ORDER and association exactly as the binary has them;
  * frames: that tree evaluated in binary32, one IEEE operation per SPIR-V operation, on whole frames at the reference's UBO
    (800 x 608: RTCHAP06/main.cpp:101-120) and BASELINE config 1 (400 x 225): CRC-32 of the RGBA8 bytes, the number of sphere
    pixels, a few hundred sampled pixels.

What SPIR-V leaves to the driver is filled in by the build's stated conventions and named in the file:
OpDot = left-to-right sum of products without contraction, Normalize(v) = v / Sqrt(Dot(v, v)), Sqrt and FDiv correctly rounded,
imageStore to rgba8 = clamp, x255, round half up, alpha = the stored 0.0.  So this pins the operation order of the one branch that has no
screenshot (CH06's normals) to the reference's binary; it does not pin the driver's precision, and `parity` stays "partial".
tests/test_oracle_golden.py compares the ORACLE's frames with these and checks the associations the oracle restates.
"""
import json
import os
import struct
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
SHADERS = {"CH06": "RTCHAP06/Shaders/raytrace06.comp.spv", "CH05": "RTCHAP05/RTCHAP05/Shaders/raytrace05.comp.spv"}

OPN = {5: "Name", 6: "MemberName", 11: "ExtInstImport", 12: "ExtInst", 15: "EntryPoint", 16: "ExecutionMode", 19: "TypeVoid", 20: "TypeBool",
       21: "TypeInt", 22: "TypeFloat", 23: "TypeVector", 25: "TypeImage", 30: "TypeStruct", 32: "TypePointer", 33: "TypeFunction",
       41: "ConstantTrue", 42: "ConstantFalse", 43: "Constant", 44: "ConstantComposite", 54: "Function", 55: "FunctionParameter",
       56: "FunctionEnd", 57: "FunctionCall", 59: "Variable", 61: "Load", 62: "Store", 65: "AccessChain", 71: "Decorate", 72: "MemberDecorate",
       79: "VectorShuffle", 80: "CompositeConstruct", 81: "CompositeExtract", 99: "ImageWrite", 112: "ConvertUToF", 111: "ConvertSToF",
       124: "Bitcast", 127: "FNegate", 129: "FAdd", 131: "FSub", 133: "FMul", 136: "FDiv", 142: "VectorTimesScalar", 148: "Dot",
       169: "Select", 180: "FOrdEqual", 184: "FOrdLessThan", 186: "FOrdGreaterThan", 188: "FOrdLessThanEqual", 190: "FOrdGreaterThanEqual",
       245: "Phi", 246: "LoopMerge", 247: "SelectionMerge", 248: "Label", 249: "Branch", 250: "BranchConditional", 253: "Return", 254: "ReturnValue"}
GLSL450 = {31: "Sqrt", 69: "Normalize", 66: "Length", 46: "FMix", 43: "FClamp"}
ARITH = {"ConvertUToF", "ConvertSToF", "FNegate", "FAdd", "FSub", "FMul", "FDiv", "VectorTimesScalar", "Dot", "ExtInst", "FOrdLessThan",
         "FOrdGreaterThan", "FOrdLessThanEqual", "FOrdGreaterThanEqual", "FOrdEqual", "Select", "CompositeConstruct", "CompositeExtract",
         "VectorShuffle", "FunctionCall", "ImageWrite", "BranchConditional", "ReturnValue", "Return"}


def parse(path):
    d = open(path, "rb").read()
    w = struct.unpack("<%dI" % (len(d) // 4), d)
    assert w[0] == 0x07230203
    ins, i = [], 5
    while i < len(w):
        n, op = w[i] >> 16, w[i] & 0xFFFF
        ins.append((OPN.get(op, "Op%d" % op), list(w[i + 1:i + n])))
        i += n
    return w[1], ins


def literal_string(words):
    b = b"".join(struct.pack("<I", x) for x in words)
    return b.split(b"\0")[0].decode()


class Module:
    def __init__(self, ins):
        self.names, self.types, self.consts, self.vars, self.funcs, self.deco = {}, {}, {}, {}, {}, {}
        self.local_size = None
        cur = None
        for op, a in ins:
            if op == "Name":
                self.names[a[0]] = literal_string(a[1:])
            elif op == "ExecutionMode" and a[1] == 17:
                self.local_size = a[2:5]
            elif op == "Decorate":
                self.deco.setdefault(a[0], []).append(a[1:])
            elif op.startswith("Type"):
                self.types[a[0]] = (op, a[1:])
            elif op == "Constant":
                t = self.types[a[0]]
                self.consts[a[1]] = struct.unpack("<f", struct.pack("<I", a[2]))[0] if t[0] == "TypeFloat" else ("int", a[2])
            elif op == "ConstantComposite":
                self.consts[a[1]] = ("composite", a[2:])
            elif op == "Variable" and cur is None:
                self.vars[a[1]] = a[2]  # storage class
            elif op == "Function":
                cur = {"id": a[1], "params": [], "body": []}
                self.funcs[a[1]] = cur
            elif op == "FunctionParameter":
                cur["params"].append(a[1])
            elif op == "FunctionEnd":
                cur = None
            elif cur is not None:
                cur["body"].append((op, a))

    def fname(self, fid):
        return self.names.get(fid, "f%d" % fid).split("(")[0]


# ---- symbolic execution: scalars are expression tuples, vectors / structs are Python lists of values ----
class Ptr:
    def __init__(self, root, path=()):
        self.root, self.path = root, tuple(path)


def get_in(v, path):
    for k in path:
        v = v[k]
    return v


def set_in(v, path, x):
    if not path:
        return x
    v = list(v)
    v[path[0]] = set_in(v[path[0]], path[1:], x)
    return v


def vmap(f, *vs):
    if isinstance(vs[0], list):
        return [vmap(f, *[v[k] if isinstance(v, list) else v for v in vs]) for k in range(len(vs[0]))]
    return f(*vs)


def select(c, a, b):
    if isinstance(a, list):
        return [select(c, x, y) for x, y in zip(a, b)]
    return a if a == b else ("Select", c, a, b)


class Exec:
    def __init__(self, mod):
        self.m = mod

    def const(self, cid):
        c = self.m.consts[cid]
        if isinstance(c, tuple) and c[0] == "composite":
            return [self.const(x) for x in c[1]]
        if isinstance(c, tuple):
            return ("int", c[1])
        return ("const", c)

    def call(self, fid, args):
        f = self.m.funcs[fid]
        env = dict(zip(f["params"], args))
        labels = {a[0]: i for i, (op, a) in enumerate(f["body"]) if op == "Label"}
        return self.run(f, 0, env, {}, labels)

    def run(self, f, pc, env, mem, labels):
        """executes from pc to a return; returns (value or None, list of side effects)"""
        env, mem = dict(env), dict(mem)
        body = f["body"]
        effects = []

        def val(i):
            return env[i] if i in env else self.const(i)

        while True:
            op, a = body[pc]
            pc += 1
            if op in ("Label", "SelectionMerge"):
                continue
            if op == "Variable":
                env[a[1]] = Ptr(("local", a[1]))
                mem[("local", a[1])] = None
            elif op == "AccessChain":
                base = env[a[2]] if a[2] in env else Ptr(("global", a[2]))
                idx = [self.m.consts[i][1] for i in a[3:]]
                env[a[1]] = Ptr(base.root, base.path + tuple(idx))
            elif op == "Load":
                p = env[a[2]] if a[2] in env else Ptr(("global", a[2]))
                if p.root[0] == "global":
                    nm = self.m.names.get(p.root[1], "") or "ubo"
                    if self.m.vars.get(p.root[1]) == 1:  # Input: gl_GlobalInvocationID
                        env[a[1]] = [("gid", 0), ("gid", 1), ("gid", 2)] if not p.path else ("gid", p.path[0])
                    elif self.m.vars.get(p.root[1]) == 2:  # Uniform: the UBO block
                        env[a[1]] = ("ubo", p.path[-1])
                    else:
                        env[a[1]] = ("image", nm)
                else:
                    env[a[1]] = get_in(mem[p.root], p.path)
            elif op == "Store":
                p = env[a[0]]
                mem[p.root] = set_in(mem[p.root] if mem[p.root] is not None else self.blank(p.root), p.path, val(a[1]))
            elif op == "FunctionCall":
                # arguments are pointers to locals (glslang passes by reference): the callee gets their current values in fresh cells
                cargs = []
                cmem_env = {}
                callee = self.m.funcs[a[2]]
                for pid, arg in zip(callee["params"], a[3:]):
                    p = env[arg]
                    cargs.append(("arg", get_in(mem[p.root], p.path)))
                r, _ = self.call_with_values(a[2], [c[1] for c in cargs])
                env[a[1]] = r
            elif op == "ExtInst":
                name = GLSL450.get(a[3], "Ext%d" % a[3])
                x = val(a[4])
                env[a[1]] = ("Normalize", x) if name == "Normalize" else vmap(lambda s: (name, s), x)
                if name == "Normalize":
                    env[a[1]] = [("NormalizeComponent", x, k) for k in range(len(x))]
            elif op in ("FAdd", "FSub", "FMul", "FDiv"):
                env[a[1]] = vmap(lambda x, y: (op, x, y), val(a[2]), val(a[3]))
            elif op == "VectorTimesScalar":
                s = val(a[3])
                env[a[1]] = [("FMul", x, s) for x in val(a[2])]
            elif op == "FNegate":
                env[a[1]] = vmap(lambda x: ("FNegate", x), val(a[2]))
            elif op == "Dot":
                env[a[1]] = ("Dot", val(a[2]), val(a[3]))
            elif op in ("ConvertUToF", "ConvertSToF"):
                env[a[1]] = vmap(lambda x: (op, x), val(a[2]))
            elif op == "Bitcast":
                env[a[1]] = val(a[2])
            elif op == "CompositeConstruct":
                out = []
                for x in a[2:]:
                    v = val(x)
                    out += v if isinstance(v, list) else [v]
                env[a[1]] = out
            elif op == "CompositeExtract":
                env[a[1]] = get_in(val(a[2]), a[3:])
            elif op == "VectorShuffle":
                both = val(a[2]) + val(a[3])
                env[a[1]] = [both[k] for k in a[4:]]
            elif op in ("FOrdLessThan", "FOrdGreaterThan", "FOrdLessThanEqual", "FOrdGreaterThanEqual", "FOrdEqual"):
                env[a[1]] = (op, val(a[2]), val(a[3]))
            elif op == "ImageWrite":
                effects.append(("ImageWrite", val(a[1]), val(a[2])))
            elif op == "Branch":
                pc = labels[a[0]]
            elif op == "BranchConditional":
                c = val(a[0])
                rt, et = self.run(f, labels[a[1]], env, mem, labels)
                rf, ef = self.run(f, labels[a[2]], env, mem, labels)
                assert len(et) == len(ef)
                eff = effects + [(x[0],) + tuple(select(c, p, q) for p, q in zip(x[1:], y[1:])) for x, y in zip(et, ef)]
                return (select(c, rt, rf) if rt is not None else None), eff
            elif op == "ReturnValue":
                return val(a[0]), effects
            elif op == "Return":
                return None, effects
            else:
                raise NotImplementedError(op)

    def blank(self, root):
        return [[None] * 4 for _ in range(4)]  # (a struct / vector cell written member by member)

    def call_with_values(self, fid, values):
        f = self.m.funcs[fid]
        env, mem = {}, {}
        for pid, v in zip(f["params"], values):
            env[pid] = Ptr(("param", pid))
            mem[("param", pid)] = v
        labels = {a[0]: i for i, (op, a) in enumerate(f["body"]) if op == "Label"}
        return self.run(f, 0, env, mem, labels)


class Dag:
    """Expression trees as a table of nodes (a shared subexpression is stored once): node = [op, operand ...] where an operand is a
    node number, and leaves are ["const", value], ["gid", k], ["ubo", member], ["param", name(, component)]; "vec" groups components."""

    def __init__(self):
        self.nodes, self.index = [], {}

    def add(self, e):
        if isinstance(e, list):
            node = ("vec",) + tuple(self.add(x) for x in e)
        elif e[0] == "const":
            node = ("const", float(e[1]))
        elif e[0] in ("int", "gid", "ubo", "param"):
            node = tuple(e)
        elif e[0] == "NormalizeComponent":
            node = ("NormalizeComponent", self.add(e[1]), e[2])
        else:
            node = (e[0],) + tuple(self.add(x) for x in e[1:])
        if node not in self.index:
            self.index[node] = len(self.nodes)
            self.nodes.append(list(node))
        return self.index[node]


# ---- binary32 evaluation, one IEEE operation per node ----
def evaluate(e, gx, gy, ubo, cache):
    f32 = np.float32
    key = id(e)
    if key in cache:
        return cache[key]
    if isinstance(e, list):
        r = [evaluate(x, gx, gy, ubo, cache) for x in e]
    else:
        op = e[0]
        ev = lambda x: evaluate(x, gx, gy, ubo, cache)
        if op == "const":
            r = f32(e[1])
        elif op == "gid":
            r = (gx, gy, np.uint32(0))[e[1]]
        elif op == "ubo":
            r = f32(ubo[e[1]])
        elif op == "ConvertUToF":
            r = ev(e[1]).astype(f32)
        elif op == "FAdd":
            r = (ev(e[1]) + ev(e[2])).astype(f32)
        elif op == "FSub":
            r = (ev(e[1]) - ev(e[2])).astype(f32)
        elif op == "FMul":
            r = (ev(e[1]) * ev(e[2])).astype(f32)
        elif op == "FDiv":
            r = (ev(e[1]) / ev(e[2])).astype(f32)
        elif op == "FNegate":
            r = -ev(e[1])
        elif op == "Sqrt":
            r = np.sqrt(ev(e[1])).astype(f32)
        elif op == "Dot":  # convention: ((x0 y0 + x1 y1) + x2 y2), no contraction
            x, y = ev(e[1]), ev(e[2])
            acc = (x[0] * y[0]).astype(f32)
            for k in range(1, len(x)):
                acc = (acc + (x[k] * y[k]).astype(f32)).astype(f32)
            r = acc
        elif op == "NormalizeComponent":  # convention: v / Sqrt(Dot(v, v))
            v = ev(e[1])
            acc = (v[0] * v[0]).astype(f32)
            for k in range(1, len(v)):
                acc = (acc + (v[k] * v[k]).astype(f32)).astype(f32)
            r = (v[e[2]] / np.sqrt(acc).astype(f32)).astype(f32)
        elif op == "FOrdLessThan":
            r = ev(e[1]) < ev(e[2])
        elif op == "FOrdGreaterThan":
            r = ev(e[1]) > ev(e[2])
        elif op == "Select":
            r = np.where(ev(e[1]), ev(e[2]), ev(e[3]))
        else:
            raise NotImplementedError(op)
    cache[key] = r
    return r


def render(tree, ubo):
    W, H = int(ubo[0]), int(ubo[1])  # RTCHAP06/main.cpp:106: the image extent is the float size truncated
    gy, gx = np.meshgrid(np.arange(H, dtype=np.uint32), np.arange(W, dtype=np.uint32), indexing="ij")
    with np.errstate(all="ignore"):
        col = evaluate(tree, gx, gy, ubo, {})
    out = np.zeros((H, W, 4), np.uint8)
    for k in range(4):  # convention: rgba8 imageStore = clamp to [0, 1], x 255, round half up
        c = np.broadcast_to(np.asarray(col[k], np.float32), (H, W))
        c = np.where(c > 0, np.where(c < 1, c, np.float32(1)), np.float32(0)).astype(np.float32)
        out[..., k] = ((c * np.float32(255)).astype(np.float32) + np.float32(0.5)).astype(np.float32).astype(np.int32)
    return out


def ubo_for(w, h):  # RTCHAP06/main.cpp:102-120 in binary32
    f = np.float32
    aspect = f(f(w) / f(h))
    return [f(w), f(f(w) / aspect), f(2.0), f(f(2.0) / aspect), f(1.0)]


def main():
    out = {"_what": __doc__.split("\n\n")[0], "conventions": {
        "Dot": "((x0*y0 + x1*y1) + x2*y2), each product and sum rounded to binary32, no contraction",
        "Normalize": "v / Sqrt(Dot(v, v)) componentwise", "Sqrt_FDiv": "correctly rounded", "imageStore_rgba8": "clamp[0,1], *255, +0.5, truncate; alpha from the stored 0.0"},
        "shaders": {}}
    for mode, rel in SHADERS.items():
        version, ins = parse(os.path.join(REF, rel))
        m = Module(ins)
        entry = next(a[1] for op, a in ins if op == "EntryPoint")
        info = {"file": rel, "spirv_version": "%d.%d" % ((version >> 16) & 255, (version >> 8) & 255), "local_size": m.local_size,
                "float_constants": sorted({float(c) for c in m.consts.values() if isinstance(c, float)}),
                "no_contraction_decorations": sum(1 for v in m.deco.values() for d in v if d[0] == 42), "functions": {}}
        ex = Exec(m)
        dag = Dag()
        for fid, f in m.funcs.items():
            ops = []
            for op, a in f["body"]:
                if op == "ExtInst":
                    ops.append("ExtInst:" + GLSL450.get(a[3], str(a[3])))
                elif op == "FunctionCall":
                    ops.append("Call:" + m.fname(a[2]))
                elif op in ARITH:
                    ops.append(op)
            info["functions"][m.fname(fid)] = {"ops": ops, "ext_insts": sorted({o.split(":")[1] for o in ops if o.startswith("ExtInst:")})}
        # expression trees: each helper over symbolic parameters, main over gid / ubo
        for fid, f in m.funcs.items():
            nm = m.fname(fid)
            if fid == entry:
                _, eff = ex.call_with_values(fid, [])
                (kind, coord, colour), = eff
                info["functions"][nm]["stores"] = {"coord": dag.add(coord), "colour": dag.add(colour)}
                tree = colour
            else:
                params = []
                for k, pid in enumerate(f["params"]):
                    pn = m.names.get(pid, "p%d" % k)
                    t = m.types[m.types[next(a[0] for op, a in ins if op == "FunctionParameter" and a[1] == pid)][1][1]]
                    if t[0] == "TypeStruct":  # Ray {vec3 orig, vec3 dir}
                        params.append([[("param", pn + ".orig", c) for c in range(3)], [("param", pn + ".dir", c) for c in range(3)]])
                    elif t[0] == "TypeVector":
                        params.append([("param", pn, c) for c in range(t[1][1])])
                    else:
                        params.append(("param", pn))
                r, _ = ex.call_with_values(fid, params)
                info["functions"][nm]["returns"] = dag.add(r)
        frames = {}
        for (w, h) in ((800, 608), (400, 225)):
            ubo = ubo_for(w, h)
            img = render(tree, ubo)
            ys = np.linspace(0, h - 1, 12).astype(int)
            xs = np.linspace(0, w - 1, 16).astype(int)
            frames["%dx%d" % (w, h)] = {"ubo": [float(x) for x in ubo], "crc32_rgba8_row0_bottom": zlib.crc32(img.tobytes()) & 0xFFFFFFFF,
                                        "alpha_max": int(img[..., 3].max()),
                                        "samples": [[int(x), int(y)] + img[y, x, :3].tolist() for y in ys for x in xs]}
        info["expression_nodes"] = dag.nodes
        info["frames"] = frames
        out["shaders"][mode] = info
    json.dump(out, open(os.path.join(HERE, "spirv_witness.json"), "w"), indent=1)
    print("wrote spirv_witness.json:", {k: {f: v["frames"][f]["crc32_rgba8_row0_bottom"] for f in v["frames"]} for k, v in out["shaders"].items()})


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("build container only: /root/reference is not here")
    main()
