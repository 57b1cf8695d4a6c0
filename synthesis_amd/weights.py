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


# ---- the `.ot` VarStore container (synthesis/src/alpha_zero.rs:37,97 `vs.save(model_i.ot)`, reloaded at :194) --------------------
# tch's VarStore::save hands the named variables to libtorch's torch::serialize::OutputArchive: a zip archive (stored, not deflated)
# with one top-level directory holding `data.pkl` — a protocol-2 pickle of a `__torch__.Module` object whose state maps variable names
# (`l_1.weight` ... `l_5.bias`, study-connect4/src/policies.rs:20-24) to tensors rebuilt from `data/<key>` (raw little-endian storages)
# — plus `code/__torch__.py` (the parameter declarations), `constants.pkl`, `version`. Read here with `zipfile` and an unpickler that
# admits exactly the globals such an archive needs: no torch at run time, nothing executable is ever looked up.
# No `.ot` written by the reference itself exists in this build (no Rust toolchain): the layout is pinned by archives the image's
# libtorch writes through that same OutputArchive (tests/golden/make_ot_golden.py).
import collections
import io
import pickle
import struct
import zipfile

_STORAGE_DTYPES = {"FloatStorage": np.dtype("<f4"), "DoubleStorage": np.dtype("<f8"), "HalfStorage": np.dtype("<f2"),
                   "LongStorage": np.dtype("<i8"), "IntStorage": np.dtype("<i4")}


class _StorageType:
    def __init__(self, name):
        self.dtype = _STORAGE_DTYPES[name]


class _JitObject:
    """stand-in for the archive's `__torch__.Module`: keeps the state dictionary BUILD hands it"""
    def __init__(self, *a):
        self.state = {}

    def __setstate__(self, state):
        self.state = state


def _rebuild_tensor_v2(storage, offset, size, stride, requires_grad=False, backward_hooks=None, metadata=None):
    size, stride = tuple(int(v) for v in size), tuple(int(v) for v in stride)
    offset = int(offset)
    # the numbers come from the archive's pickle: every element the view can reach must lie inside the storage
    if len(size) != len(stride) or offset < 0 or any(v < 0 for v in size) or any(v < 0 for v in stride):
        raise ValueError(f"tensor with offset {offset}, size {size}, stride {stride}: not a view of its storage")
    if all(v > 0 for v in size) and offset + sum((n - 1) * st for n, st in zip(size, stride)) >= storage.size:
        raise ValueError(f"tensor with offset {offset}, size {size}, stride {stride} reaches past its storage of {storage.size} elements")
    if len(size) == 0:
        return storage[offset:offset + 1].reshape(()).copy()
    if any(v == 0 for v in size):
        return np.zeros(size, storage.dtype)
    return np.lib.stride_tricks.as_strided(storage[offset:], shape=size, strides=tuple(s * storage.itemsize for s in stride)).copy()


class _OtUnpickler(pickle.Unpickler):
    def __init__(self, data, read_record):
        super().__init__(io.BytesIO(data))
        self._read_record = read_record

    def find_class(self, module, name):
        if module == "__torch__" or module.startswith("__torch__."):
            return _JitObject
        if module == "torch._utils" and name == "_rebuild_tensor_v2":
            return _rebuild_tensor_v2
        if module == "torch._utils" and name == "_rebuild_parameter":
            return lambda data, requires_grad=True, backward_hooks=None: data
        if module == "torch" and name in _STORAGE_DTYPES:
            return _StorageType(name)
        if module == "collections" and name == "OrderedDict":
            return collections.OrderedDict
        raise pickle.UnpicklingError(f"a VarStore archive has no business with {module}.{name}")

    def persistent_load(self, pid):
        if not (isinstance(pid, tuple) and len(pid) >= 5 and pid[0] == "storage" and isinstance(pid[1], _StorageType)):
            raise pickle.UnpicklingError(f"unexpected persistent id {pid!r}")
        raw = self._read_record("data/" + str(pid[2]))
        arr = np.frombuffer(raw, dtype=pid[1].dtype)
        if arr.size < int(pid[4]):
            raise ValueError(f"storage {pid[2]} holds {arr.size} elements, the archive says {pid[4]}")
        return arr


def load_ot_tensors(path):
    """name -> ndarray for every variable of a VarStore archive (`.ot`)."""
    with zipfile.ZipFile(path) as z:
        pkl = [n for n in z.namelist() if n.endswith("/data.pkl") and n.count("/") == 1]
        if len(pkl) != 1:
            raise ValueError(f"{path}: not a libtorch archive (no single <root>/data.pkl)")
        root = pkl[0][: -len("data.pkl")]
        order = (z.read(root + "byteorder").decode().strip() if root + "byteorder" in z.namelist() else "little")
        if order != "little":
            raise ValueError(f"{path}: {order}-endian archives are not supported")
        obj = _OtUnpickler(z.read(pkl[0]), lambda rec: z.read(root + rec)).load()
    state = obj.state if isinstance(obj, _JitObject) else obj
    if not isinstance(state, dict):
        raise ValueError(f"{path}: the archive's root object carries no variables")
    # tch writes a VarStore through Tensor::save_multi, which stores every name with '.' replaced by '|' (a TorchScript attribute
    # name cannot hold a dot) and maps it back in load_multi: `l_1|weight` on disk is the variable `l_1.weight`
    return {str(k).replace("|", "."): np.asarray(v) for k, v in state.items() if isinstance(v, np.ndarray)}


def load_ot(path):
    """The flat f32 blob (`l_1.weight, l_1.bias, ... l_5.bias`; syn_load_weights' order) of a Connect4Net VarStore archive."""
    t = load_ot_tensors(path)
    parts = []
    for l in range(5):
        w, b = t.get(f"l_{l + 1}.weight"), t.get(f"l_{l + 1}.bias")
        if w is None or b is None:
            raise ValueError(f"{path}: no l_{l + 1}.weight / l_{l + 1}.bias (variables: {sorted(t)})")
        if w.shape != (DIMS[l + 1], DIMS[l]) or b.shape != (DIMS[l + 1],):
            raise ValueError(f"{path}: l_{l + 1} has shapes {w.shape} / {b.shape}, Connect4Net needs {(DIMS[l + 1], DIMS[l])} / {(DIMS[l + 1],)}")
        parts += [w.astype(np.float32).ravel(), b.astype(np.float32).ravel()]
    return np.concatenate(parts)


def _pickle_varstore(named):
    """data.pkl of an OutputArchive holding the f32 tensors `named` [(name, array)]: the opcode stream libtorch's pickler emits"""
    out = bytearray(b"\x80\x02c__torch__\nModule\nq\x00)\x81}(")
    memo = 1

    def put():
        nonlocal memo
        memo += 1
        return b"q" + bytes([memo - 1]) if memo - 1 < 256 else b"r" + struct.pack("<I", memo - 1)

    def uni(sv):
        b = sv.encode()
        return b"X" + struct.pack("<I", len(b)) + b

    def integer(v):
        return b"K" + bytes([v]) if 0 <= v < 256 else (b"M" + struct.pack("<H", v) if v < 65536 else b"J" + struct.pack("<i", v))

    first = True
    for key, (name, arr) in enumerate(named):
        out += uni(name) + put()
        if first:
            out += b"ctorch._utils\n_rebuild_tensor_v2\n" + put()
            rebuild = memo - 1
        else:
            out += b"h" + bytes([rebuild])
        out += b"(("
        if first:
            out += uni("storage") + put(); m_storage = memo - 1
            out += b"ctorch\nFloatStorage\n" + put(); m_float = memo - 1
        else:
            out += b"h" + bytes([m_storage]) + b"h" + bytes([m_float])
        out += uni(str(key)) + put()
        if first:
            out += uni("cpu") + put(); m_cpu = memo - 1
        else:
            out += b"h" + bytes([m_cpu])
        out += integer(arr.size) + b"tQ" + put() + integer(0)
        out += b"(" + b"".join(integer(d) for d in arr.shape) + b"t"
        strides = [int(np.prod(arr.shape[i + 1:])) for i in range(arr.ndim)]
        out += b"(" + b"".join(integer(d) for d in strides) + b"t" + b"\x89"
        if first:
            out += b"ccollections\nOrderedDict\n" + put(); m_od = memo - 1
        else:
            out += b"h" + bytes([m_od])
        out += b")RtR" + put()
        first = False
    out += b"ub" + put() + b"."
    return bytes(out)


def save_ot(blob, path, root="archive", names="tch"):
    """Writes the flat Connect4Net blob as a VarStore archive `vs.load(path)` / torch.jit.load read: the inverse of load_ot.
    names="tch": the on-disk names `vs.save` produces (`l_1|weight`: tch's save_multi replaces '.' by '|', load_multi maps it
    back); names="dotted": `l_1.weight`, what a bare OutputArchive::write(name, ...) stores."""
    if names not in ("tch", "dotted"):
        raise ValueError("names must be 'tch' or 'dotted'")
    blob = np.ascontiguousarray(blob, dtype=np.float32).ravel()
    if blob.size != NUM_PARAMS:
        raise ValueError(f"Connect4Net has {NUM_PARAMS} parameters, got {blob.size}")
    named, off = [], 0
    for l in range(5):
        w = blob[off:off + DIMS[l] * DIMS[l + 1]].reshape(DIMS[l + 1], DIMS[l]); off += w.size
        b = blob[off:off + DIMS[l + 1]]; off += b.size
        sep = "|" if names == "tch" else "."
        named += [(f"l_{l + 1}{sep}weight", w), (f"l_{l + 1}{sep}bias", b)]
    names = "".join(f'"{n}", ' for n, _ in named)
    code = ("class Module(Module):\n  __parameters__ = [" + names + "]\n  __buffers__ = []\n  __annotations__ = []\n" +
            "".join(f'  __annotations__["{n}"] = Tensor\n' for n, _ in named))
    with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as z:
        for key, (_, arr) in enumerate(named):
            z.writestr(f"{root}/data/{key}", np.ascontiguousarray(arr, "<f4").tobytes())
        z.writestr(f"{root}/data.pkl", _pickle_varstore(named))
        z.writestr(f"{root}/code/__torch__.py", code)
        z.writestr(f"{root}/constants.pkl", b"\x80\x02).")
        z.writestr(f"{root}/version", b"3\n")
        z.writestr(f"{root}/byteorder", b"little")
