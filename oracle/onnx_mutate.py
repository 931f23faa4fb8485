"""TEST INFRASTRUCTURE -- rewrites a voice .onnx the way post-export tooling does, to harden the structural loader
(piper_amd/csrc/onnx_reader.cpp) against what real voices look like (SURVEY.md section 7 hard part A; the reference
recommends onnx-simplifier after export, TRAINING.md:234): initialisers renamed to bare numerals, node names stripped,
weights moved into Constant nodes, identical tensors de-duplicated, raw_data re-encoded as float_data.

No `onnx` package: a minimal protobuf reader / writer for the fields of ModelProto / GraphProto / NodeProto /
TensorProto that matter (everything else is copied through byte for byte).
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

Field = Tuple[int, int, object]       # (field number, wire type, value: int | bytes)


def _varint(buf: bytes, i: int) -> Tuple[int, int]:
    r, sh = 0, 0
    while True:
        c = buf[i]
        i += 1
        r |= (c & 0x7F) << sh
        if not c & 0x80:
            return r, i
        sh += 7


def parse(buf: bytes) -> List[Field]:
    out, i = [], 0
    while i < len(buf):
        key, i = _varint(buf, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 1:
            v, i = buf[i:i + 8], i + 8
        elif wt == 2:
            n, i = _varint(buf, i)
            v, i = buf[i:i + n], i + n
        elif wt == 5:
            v, i = buf[i:i + 4], i + 4
        else:
            raise ValueError(f"wire type {wt}")
        out.append((f, wt, v))
    return out


def _enc_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def serialize(fields: List[Field]) -> bytes:
    out = bytearray()
    for f, wt, v in fields:
        out += _enc_varint((f << 3) | wt)
        if wt == 0:
            out += _enc_varint(v)
        elif wt == 2:
            out += _enc_varint(len(v)) + v
        else:
            out += v
    return bytes(out)


def _get(fields: List[Field], num: int):
    return [v for f, _, v in fields if f == num]


class Tensor:
    def __init__(self, blob: bytes):
        self.fields = parse(blob)

    @property
    def name(self) -> str:
        v = _get(self.fields, 8)
        return v[0].decode() if v else ""

    @name.setter
    def name(self, n: str):
        self.fields = [(f, wt, v) for f, wt, v in self.fields if f != 8] + [(8, 2, n.encode())]

    @property
    def dtype(self) -> int:
        return _get(self.fields, 2)[0]

    def key(self) -> bytes:
        """content identity: dims + dtype + payload"""
        return serialize([(f, wt, v) for f, wt, v in self.fields if f != 8])

    def to_float_data(self):
        """raw_data (field 9) -> packed float_data (field 4), float tensors only"""
        raw = _get(self.fields, 9)
        if self.dtype != 1 or not raw:
            return
        self.fields = [(f, wt, v) for f, wt, v in self.fields if f != 9] + [(4, 2, raw[0])]

    def set_constant(self, value: float):
        raw = _get(self.fields, 9)
        n = len(raw[0]) // 4
        self.fields = [(f, wt, v) for f, wt, v in self.fields if f != 9] + [(9, 2, struct.pack("<f", value) * n)]

    def blob(self) -> bytes:
        return serialize(self.fields)


class Model:
    """ModelProto with its GraphProto opened: nodes (field 1) and initialisers (field 5) editable."""

    def __init__(self, data: bytes):
        self.model = parse(data)
        gi = [i for i, (f, wt, _) in enumerate(self.model) if f == 7 and wt == 2]
        self.gi = gi[0]
        self.graph = parse(self.model[self.gi][2])
        self.inits = [Tensor(v) for f, _, v in self.graph if f == 5]
        self.nodes = [parse(v) for f, _, v in self.graph if f == 1]
        self.rest = [(f, wt, v) for f, wt, v in self.graph if f not in (1, 5)]

    def save(self) -> bytes:
        g = [(1, 2, serialize(n)) for n in self.nodes] + [(5, 2, t.blob()) for t in self.inits] + self.rest
        m = list(self.model)
        m[self.gi] = (7, 2, serialize(g))
        return serialize(m)

    # ---- mutations
    def _rename_values(self, mapping: Dict[str, str]):
        for n in self.nodes:
            for i, (f, wt, v) in enumerate(n):
                if f in (1, 2) and wt == 2 and v.decode() in mapping:
                    n[i] = (f, wt, mapping[v.decode()].encode())
        # graph inputs / value_info that mention an initialiser (old exporters list them as inputs)
        rest = []
        for f, wt, v in self.rest:
            if f in (11, 13) and wt == 2:
                vi = parse(v)
                vi = [(ff, w, mapping.get(x.decode(), x.decode()).encode() if ff == 1 and w == 2 else x) for ff, w, x in vi]
                v = serialize(vi)
            rest.append((f, wt, v))
        self.rest = rest

    def rename_initializers_to_numerals(self, start: int = 1000):
        mapping = {}
        for k, t in enumerate(self.inits):
            mapping[t.name] = str(start + k)
            t.name = mapping[t.name]
        self._rename_values(mapping)

    def strip_node_names(self):
        self.nodes = [[(f, wt, v) for f, wt, v in n if f != 3] for n in self.nodes]

    def initializers_to_constants(self, every: int = 1):
        """Every `every`-th initialiser becomes a Constant node (attribute `value`) in front of the graph."""
        keep, consts = [], []
        for k, t in enumerate(self.inits):
            if k % every:
                keep.append(t)
                continue
            name = t.name
            tt = Tensor(t.blob())
            tt.fields = [(f, wt, v) for f, wt, v in tt.fields if f != 8]           # Constant values are anonymous
            attr = serialize([(1, 2, b"value"), (5, 2, tt.blob()), (20, 0, 4)])    # AttributeProto: name, t, type = TENSOR
            consts.append([(2, 2, name.encode()), (4, 2, b"Constant"), (5, 2, attr)])
        self.inits = keep
        self.nodes = consts + self.nodes

    def dedup(self) -> int:
        seen, mapping, keep = {}, {}, []
        for t in self.inits:
            k = t.key()
            if k in seen:
                mapping[t.name] = seen[k]
            else:
                seen[k] = t.name
                keep.append(t)
        self.inits = keep
        self._rename_values(mapping)
        return len(mapping)

    def float_data(self):
        for t in self.inits:
            t.to_float_data()


# ---- round 5: what onnx-simplifier / torch-1.x exports leave behind --------------------------------
def _attr_ints(name: str, values) -> bytes:
    """AttributeProto {name, ints, type = INTS}"""
    f = [(1, 2, name.encode())] + [(8, 0, int(v) & 0xFFFFFFFFFFFFFFFF) for v in values] + [(20, 0, 7)]
    return serialize(f)


def _int64_tensor(name: str, values) -> "Tensor":
    t = Tensor(serialize([(1, 0, len(values)), (2, 0, 7), (8, 2, name.encode()),
                          (9, 2, struct.pack("<%dq" % len(values), *values))]))
    return t


def _dims(t: "Tensor"):
    out = []
    for f, wt, v in t.fields:
        if f == 1:
            if wt == 0:
                out.append(v)
            else:
                i = 0
                while i < len(v):
                    x, i = _varint(v, i)
                    out.append(x)
    return out


def _set_dims(t: "Tensor", dims):
    t.fields = [(f, wt, v) for f, wt, v in t.fields if f != 1]
    t.fields = [(1, 0, int(d)) for d in dims] + t.fields


def _node_op(n) -> str:
    v = _get(n, 4)
    return v[0].decode() if v else ""


class Model(Model):                                                      # noqa: F811 -- extends the class above
    def _rewire(self, old: str, new: str, skip=()):
        for k, n in enumerate(self.nodes):
            if k in skip:
                continue
            for i, (f, wt, v) in enumerate(n):
                if f == 1 and wt == 2 and v.decode() == old:
                    n[i] = (f, wt, new.encode())

    def identity_shared(self, every: int = 2):
        """Every `every`-th initialiser is consumed through an Identity node (how exporters share one tensor between
        modules; onnx-simplifier keeps them when the tensor has several readers)."""
        new_nodes = []
        for k, t in enumerate(self.inits):
            if k % every:
                continue
            alias = t.name + "__id"
            self._rewire(t.name, alias)
            new_nodes.append([(1, 2, t.name.encode()), (2, 2, alias.encode()), (4, 2, b"Identity")])
        self.nodes = new_nodes + self.nodes
        return len(new_nodes)

    def unsqueeze_k1_weights(self, axes_as_input: bool = True, axes=(2,)):
        """Conv weights [out, in, 1] stored as matrices [out, in] with an Unsqueeze in front of the conv -- axes as an
        int64 input (opset >= 13) or as an attribute (older opsets)."""
        new_nodes, new_inits, n = [], [], 0
        for t in list(self.inits):
            d = _dims(t)
            if t.dtype != 1 or len(d) != 3 or d[2] != 1 or d[0] == 1:
                continue
            _set_dims(t, d[:2])
            alias = t.name + "__u"
            self._rewire(t.name, alias)
            if axes_as_input:
                ax = _int64_tensor(t.name + "__axes", list(axes))
                new_inits.append(ax)
                new_nodes.append([(1, 2, t.name.encode()), (1, 2, ax.name.encode()), (2, 2, alias.encode()), (4, 2, b"Unsqueeze")])
            else:
                new_nodes.append([(1, 2, t.name.encode()), (2, 2, alias.encode()), (4, 2, b"Unsqueeze"),
                                  (5, 2, _attr_ints("axes", list(axes)))])
            n += 1
        self.inits += new_inits
        self.nodes = new_nodes + self.nodes
        return n

    def transpose_conv_weights(self, every: int = 2):
        """Conv weights with k > 1 stored as [in, out, k] with a Transpose(perm = 1, 0, 2) in front of the conv."""
        import numpy as np
        new_nodes, n, k = [], 0, 0
        for t in self.inits:
            d = _dims(t)
            raw = _get(t.fields, 9)
            if t.dtype != 1 or len(d) != 3 or d[2] <= 1 or d[0] == 1 or not raw:
                continue
            k += 1
            if k % every:
                continue
            a = np.frombuffer(raw[0], dtype="<f4").reshape(d).transpose(1, 0, 2)
            t.fields = [(f, wt, v) for f, wt, v in t.fields if f != 9] + [(9, 2, np.ascontiguousarray(a).tobytes())]
            _set_dims(t, [d[1], d[0], d[2]])
            alias = t.name + "__t"
            self._rewire(t.name, alias)
            new_nodes.append([(1, 2, t.name.encode()), (2, 2, alias.encode()), (4, 2, b"Transpose"),
                              (5, 2, _attr_ints("perm", [1, 0, 2]))])
            n += 1
        self.nodes = new_nodes + self.nodes
        return n

    def reshape_biases(self):
        """1-D conv biases stored as [1, C] with a Squeeze(axes = 0) (input form) in front of their consumer."""
        new_nodes, new_inits, n = [], [], 0
        conv_bias = set()
        for nd in self.nodes:
            if _node_op(nd) in ("Conv", "ConvTranspose"):
                ins = [v.decode() for f, wt, v in nd if f == 1 and wt == 2]
                if len(ins) > 2:
                    conv_bias.add(ins[2])
        ax = _int64_tensor("squeeze_axes_0", [0])
        for t in self.inits:
            d = _dims(t)
            if t.name in conv_bias and t.dtype == 1 and len(d) == 1:
                _set_dims(t, [1, d[0]])
                alias = t.name + "__s"
                self._rewire(t.name, alias)
                new_nodes.append([(1, 2, t.name.encode()), (1, 2, ax.name.encode()), (2, 2, alias.encode()), (4, 2, b"Squeeze")])
                n += 1
        if n:
            self.inits.append(ax)
        self.nodes = new_nodes + self.nodes
        return n

    def make_external(self, index: int = 0) -> str:
        """Initialiser `index` (among the float ones) points at another file (data_location = EXTERNAL + external_data
        entries) and carries no payload: models > 2 GB and some tools store weights that way."""
        fl = [t for t in self.inits if t.dtype == 1 and _get(t.fields, 9)]
        t = fl[index % len(fl)]
        entry = lambda k, v: serialize([(1, 2, k.encode()), (2, 2, v.encode())])       # noqa: E731
        t.fields = [(f, wt, v) for f, wt, v in t.fields if f != 9] + \
            [(13, 2, entry("location", "weights.bin")), (13, 2, entry("offset", "0")), (14, 0, 1)]
        return t.name

    def axes_inputs_to_attributes(self) -> int:
        """Squeeze / Unsqueeze / Split / ReduceSum nodes in the opset >= 13 form (axes / split as an int64 INPUT) rewritten
        to the older attribute form -- what a file exported at a lower opset (or down-converted) looks like."""
        by_name = {t.name: t for t in self.inits}
        consts = {}
        for nd in self.nodes:                       # int64 Constant nodes
            if _node_op(nd) == "Constant":
                outs = [v.decode() for f, wt, v in nd if f == 2 and wt == 2]
                for f, wt, v in nd:
                    if f == 5 and wt == 2:
                        a = parse(v)
                        tt = _get(a, 5)
                        if tt and outs:
                            consts[outs[0]] = Tensor(tt[0])
        n = 0
        for nd in self.nodes:
            op = _node_op(nd)
            if op not in ("Squeeze", "Unsqueeze", "Split", "ReduceSum"):
                continue
            ins = [(i, v.decode()) for i, (f, wt, v) in enumerate(nd) if f == 1 and wt == 2]
            if len(ins) < 2:
                continue
            pos, name = ins[1]
            t = by_name.get(name) or consts.get(name)
            if t is None or t.dtype != 7:
                continue
            raw = _get(t.fields, 9)
            if not raw:
                continue
            vals = struct.unpack("<%dq" % (len(raw[0]) // 8), raw[0])
            del nd[pos]
            nd.append((5, 2, _attr_ints("split" if op == "Split" else "axes", vals)))
            n += 1
        return n

    def shuffle_nodes(self, seed: int, keep_conv_order: bool = False):
        """Another valid topological order of the graph (ONNX requires no particular one). keep_conv_order: the
        convolutions other than an attention layer's q / k / v keep their relative order (only the element-wise glue and
        the parallel q / k / v branches move)."""
        import random
        rng = random.Random(seed)
        outs = [[v.decode() for f, wt, v in nd if f == 2 and wt == 2] for nd in self.nodes]
        ins = [[v.decode() for f, wt, v in nd if f == 1 and wt == 2 and v] for nd in self.nodes]
        producer = {o: i for i, os_ in enumerate(outs) for o in os_}
        deps = [set(producer[x] for x in ins[i] if x in producer) for i in range(len(self.nodes))]
        is_conv = [_node_op(nd) in ("Conv", "ConvTranspose") for nd in self.nodes]
        free_conv = set()
        if keep_conv_order:
            # q / k / v: three consecutive convs (in file order) that read the same input value
            cidx = [i for i, c in enumerate(is_conv) if c]
            for a, b, c in zip(cidx, cidx[1:], cidx[2:]):
                if ins[a][:1] == ins[b][:1] == ins[c][:1]:
                    free_conv.update((a, b, c))
        placed, order, ready = set(), [], []
        remaining = set(range(len(self.nodes)))
        next_conv = [i for i, c in enumerate(is_conv) if c and i not in free_conv]
        while remaining:
            ready = [i for i in remaining if deps[i] <= placed]
            if keep_conv_order:
                gate = next_conv[0] if next_conv else None
                ready = [i for i in ready if not is_conv[i] or i in free_conv or i == gate]
            i = rng.choice(sorted(ready))
            order.append(i)
            placed.add(i)
            remaining.discard(i)
            if next_conv and i == next_conv[0]:
                next_conv.pop(0)
        self.nodes = [self.nodes[i] for i in order]
        return order

    def swap_cond_layers(self, a: int = 0, b: int = 1):
        """Exchange the POSITIONS of two flow cond_layer convolutions (found by their bias initialiser's name,
        `flow.flows.N.enc.cond_layer.bias`): they read only the speaker embedding g, so any relative order among them is
        a valid topological order -- and they share one shape, so nothing but the graph's wiring tells them apart
        (ADVICE r5). Returns the two bias names."""
        idx = []
        for i, nd in enumerate(self.nodes):
            if _node_op(nd) != "Conv":
                continue
            ins = [v.decode() for f, wt, v in nd if f == 1 and wt == 2]
            if len(ins) >= 3 and ins[2].endswith(".enc.cond_layer.bias"):
                idx.append((i, ins[2]))
        if len(idx) <= max(a, b):
            raise ValueError("not a multi-speaker voice with that many coupling layers")
        (ia, na), (ib, nb) = idx[a], idx[b]
        self.nodes[ia], self.nodes[ib] = self.nodes[ib], self.nodes[ia]
        return na, nb

    def conv_bias_to_add(self, every: int = 2) -> int:
        """Every `every`-th convolution's bias leaves the Conv node and comes back as an Add behind it, stored [1, C, 1]
        (the un-fused spelling older exporters emit for some layers; onnx-simplifier fuses the other way, so both exist in
        the wild): Conv(x, W, b) -> Add(Conv(x, W), b[1, C, 1])."""
        by_name = {t.name: t for t in self.inits}
        out_nodes, n, k = [], 0, 0
        for nd in self.nodes:
            out_nodes.append(nd)
            if _node_op(nd) not in ("Conv", "ConvTranspose"):
                continue
            ins = [(i, v.decode()) for i, (f, wt, v) in enumerate(nd) if f == 1 and wt == 2]
            if len(ins) < 3 or ins[2][1] not in by_name:
                continue
            k += 1
            if k % every:
                continue
            bt = by_name[ins[2][1]]
            d = _dims(bt)
            if bt.dtype != 1 or len(d) != 1:
                continue
            b3 = Tensor(bt.blob())
            b3.name = bt.name + "__add"
            _set_dims(b3, [1, d[0], 1])
            self.inits.append(b3)
            del nd[ins[2][0]]                                            # the Conv keeps (x, W)
            oi = [i for i, (f, wt, v) in enumerate(nd) if f == 2 and wt == 2][0]
            out = nd[oi][2].decode()
            nd[oi] = (2, 2, (out + "__nobias").encode())
            out_nodes.append([(1, 2, (out + "__nobias").encode()), (1, 2, b3.name.encode()), (2, 2, out.encode()), (4, 2, b"Add")])
            n += 1
        self.nodes = out_nodes
        return n

    def gelu_div_to_mul(self) -> int:
        """GELU's x / sqrt(2) in front of Erf (modules.py:123 via F.gelu: Div by a scalar Constant) folded into a Mul by the
        reciprocal, the form constant folding leaves behind; and the LayerNorm variance's Pow(x, 2) spelled Mul(x, x)."""
        import numpy as np
        outs = {v.decode(): i for i, nd in enumerate(self.nodes) for f, wt, v in nd if f == 2 and wt == 2}
        erf_in = {[v.decode() for f, wt, v in nd if f == 1 and wt == 2][0] for nd in self.nodes if _node_op(nd) == "Erf"}
        n = 0
        for nd in self.nodes:
            op = _node_op(nd)
            ins = [(i, v.decode()) for i, (f, wt, v) in enumerate(nd) if f == 1 and wt == 2]
            out = [v.decode() for f, wt, v in nd if f == 2 and wt == 2]
            if op == "Div" and out and out[0] in erf_in and len(ins) == 2 and ins[1][1] in outs:
                cn = self.nodes[outs[ins[1][1]]]
                if _node_op(cn) != "Constant":
                    continue
                # Constant.value (attribute field 5 -> AttributeProto.t field 5): a scalar float
                for ai, (f, wt, v) in enumerate(cn):
                    if f != 5:
                        continue
                    attr = parse(v)
                    ti = [j for j, (ff, _, _) in enumerate(attr) if ff == 5]
                    if not ti:
                        continue
                    t = Tensor(attr[ti[0]][2])
                    raw = _get(t.fields, 9)
                    if t.dtype != 1 or not raw or len(raw[0]) != 4:
                        continue
                    c = np.frombuffer(raw[0], "<f4")[0]
                    t.fields = [(ff, w2, vv) for ff, w2, vv in t.fields if ff != 9] + [(9, 2, np.float32(1.0 / c).tobytes())]
                    attr[ti[0]] = (5, 2, t.blob())
                    cn[ai] = (5, 2, serialize(attr))
                    oi = [j for j, (ff, _, _) in enumerate(nd) if ff == 4][0]
                    nd[oi] = (4, 2, b"Mul")
                    n += 1
            elif op == "Pow" and len(ins) == 2:
                oi = [j for j, (ff, _, _) in enumerate(nd) if ff == 4][0]
                nd[oi] = (4, 2, b"Mul")
                nd[ins[1][0]] = (1, 2, ins[0][1].encode())              # Pow(x, 2) -> Mul(x, x)
                n += 1
        return n
