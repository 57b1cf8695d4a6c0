"""Weight interchange with the reference's tooling — host-side, no GPU involved.

The engine takes Connect4Net as one flat f32 blob in VarStore order `l_1.weight, l_1.bias, ..., l_5.weight, l_5.bias`
(study-connect4/src/policies.rs:20-24; weights row-major [out][in]). This module converts between that blob and

* the text file the reference's `export` binary writes (export/src/main.rs:43-92): a `PARAMETERS` array of base65536
  strings, two per layer (weight, bias) in sorted layer-name order, each the tensor's values as BIG-ENDIAN bf16
  (export/src/main.rs:8-41, round-to-nearest-even as written there);
* the strings `slimnn::load_1d / load_2d` consume (slimnn/src/loading.rs:3-39): base65536 of BIG-ENDIAN f32.
  (The two sides of the reference disagree — export writes 2 bytes per value, slimnn expects 4; `parse_export_text`
  therefore accepts either width and tells them apart by the tensor's known element count.)
* base65536 itself (base65536/src/lib.rs:26-56; the public qntm/Parkayun alphabet: byte pair (b1, b2) -> code point
  BLOCK_START[b2] + b1, a trailing single byte -> 5376 + b1).
"""
import re

import numpy as np

DIMS = [63, 128, 96, 64, 48, 12]
NUM_PARAMS = sum(DIMS[i] * DIMS[i + 1] + DIMS[i + 1] for i in range(5))  # 30,492
PADDING_BLOCK = 5376


def _block_starts():
    """The 256 block start code points of base65536 (same list as base65536/src/lib.rs:2-24), rebuilt from its runs:
    (first, count) runs of blocks 256 code points apart."""
    runs = [(13312, 25), (19968, 81), (41216, 3), (42240, 1), (67072, 1), (73728, 3), (77824, 4), (82944, 2),
            (92160, 2), (131072, 134)]
    out = []
    for first, count in runs:
        out.extend(first + 256 * i for i in range(count))
    assert len(out) == 256
    return out


BLOCK_START = _block_starts()
_BLOCK_INDEX = {v: i for i, v in enumerate(BLOCK_START)}


def base65536_encode(data: bytes) -> str:
    chars = []
    for i in range(0, len(data), 2):
        b1 = data[i]
        block = BLOCK_START[data[i + 1]] if i + 1 < len(data) else PADDING_BLOCK
        chars.append(chr(block + b1))
    return "".join(chars)


def base65536_decode(text: str) -> bytes:
    out = bytearray()
    for ch in text:
        cp = ord(ch)
        b1 = cp & 0xFF
        out.append(b1)
        block = cp - b1
        if block != PADDING_BLOCK:
            if block not in _BLOCK_INDEX:
                raise ValueError(f"U+{cp:04X} is not a base65536 code point")
            out.append(_BLOCK_INDEX[block])
    return bytes(out)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """export/src/main.rs:8-26, value for value (NaN keeps its top mantissa bits with the quiet bit set; otherwise
    round-half-to-even on bit 16)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    round_bit = np.uint64(0x8000)
    up = ((u & round_bit) != 0) & ((u & np.uint64(3 * 0x8000 - 1)) != 0)
    r = (u >> np.uint64(16)) + up.astype(np.uint64)
    r = np.where(nan, (u >> np.uint64(16)) | np.uint64(0x40), r)
    return (r & np.uint64(0xFFFF)).astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)


def tensor_to_string(t: np.ndarray, dtype: str = "bf16") -> str:
    """serialize_tensor (export/src/main.rs:28-41) for dtype 'bf16'; big-endian f32 (what slimnn::load_* reads) for 'f32'."""
    t = np.ascontiguousarray(t, dtype=np.float32).ravel()
    if dtype == "bf16":
        raw = f32_to_bf16_bits(t).astype(">u2").tobytes()
    elif dtype == "f32":
        raw = t.astype(">f4").tobytes()
    else:
        raise ValueError("dtype must be 'bf16' or 'f32'")
    return base65536_encode(raw)


def string_to_tensor(s: str, count: int) -> np.ndarray:
    """slimnn::load_* (loading.rs:3-39) when the string holds 4 bytes per value; the export binary's bf16 otherwise."""
    raw = base65536_decode(s)
    if len(raw) == 4 * count:
        return np.frombuffer(raw, dtype=">f4").astype(np.float32)
    if len(raw) == 2 * count:
        return bf16_bits_to_f32(np.frombuffer(raw, dtype=">u2").astype(np.uint16))
    raise ValueError(f"{len(raw)} bytes do not hold {count} f32 or bf16 values")


def split_blob(blob):
    """flat VarStore-order blob -> [(name, weight[out][in], bias[out])]"""
    blob = np.ascontiguousarray(blob, dtype=np.float32).ravel()
    if blob.size != NUM_PARAMS:
        raise ValueError(f"Connect4Net has {NUM_PARAMS} parameters, got {blob.size}")
    out, off = [], 0
    for i in range(5):
        k, o = DIMS[i], DIMS[i + 1]
        w = blob[off:off + o * k].reshape(o, k); off += o * k
        b = blob[off:off + o]; off += o
        out.append((f"l_{i + 1}", w, b))
    return out


def export_text(blob, dtype: str = "bf16") -> str:
    """The file `export <model.ot> <out>` writes (serialize_tensors, export/src/main.rs:43-92)."""
    layers = split_blob(blob)
    lines = []
    for i, (name, w, _) in enumerate(layers):
        lines.append(f"load_{w.ndim}d(&mut policy.{name}.weight, String::from(PARAMETERS[{2 * i}]));")
        lines.append(f"load_1d(&mut policy.{name}.bias, String::from(PARAMETERS[{2 * i + 1}]));")
    lines.append(f"const PARAMETERS: [&'static str; {2 * len(layers)}] = [")
    for i, (name, w, b) in enumerate(layers):
        lines.append(f"// {name} - {2 * i}")
        lines.append(f'"{tensor_to_string(w, dtype)}",')
        lines.append(f'"{tensor_to_string(b, dtype)}",')
    lines.append("];")
    return "\n".join(lines) + "\n"


def parse_export_text(text: str) -> np.ndarray:
    """PARAMETERS file -> flat f32 blob in VarStore order (what Engine.load_weights takes)."""
    body = text[text.index("const PARAMETERS"):]
    strings = re.findall(r'^"([^"]*)",\s*$', body, flags=re.M)
    if len(strings) != 10:
        raise ValueError(f"expected 10 parameter strings (5 layers x weight, bias), found {len(strings)}")
    parts = []
    for i in range(5):
        k, o = DIMS[i], DIMS[i + 1]
        parts.append(string_to_tensor(strings[2 * i], o * k))
        parts.append(string_to_tensor(strings[2 * i + 1], o))
    return np.concatenate(parts).astype(np.float32)
