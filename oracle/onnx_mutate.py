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
