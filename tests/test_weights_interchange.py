"""Weight interchange: base65536 (base65536/src/lib.rs), the `export` binary's PARAMETERS file
(export/src/main.rs:8-92) and the strings slimnn::load_* consume (slimnn/src/loading.rs:3-39)."""
import os

import numpy as np
import pytest

from synthesis_amd import weights as W


def test_base65536_known_answers_and_round_trips():
    # the reference's own test (base65536/src/lib.rs:63-69) is a round trip of b"Hello World"
    assert W.base65536_decode(W.base65536_encode(b"Hello World")) == b"Hello World"
    # published vector of the alphabet this crate implements (qntm/base65536 README)
    assert W.base65536_encode(b"hello world") == "驨ꍬ啯\U00012077ꍲᕤ"
    assert W.base65536_decode("驨ꍬ啯\U00012077ꍲᕤ") == b"hello world"
    # one code point per byte pair, a final odd byte lands in the padding block (5376 + b)
    assert W.base65536_encode(b"") == "" and W.base65536_decode("") == b""
    assert [ord(c) for c in W.base65536_encode(bytes([7]))] == [5376 + 7]
    assert [ord(c) for c in W.base65536_encode(bytes([1, 0, 2]))] == [13312 + 1, 5376 + 2]
    allpairs = bytes(v for b2 in range(256) for b1 in range(256) for v in (b1, b2))
    enc = W.base65536_encode(allpairs)
    assert len(enc) == 65536 and len(set(enc)) == 65536
    assert W.base65536_decode(enc) == allpairs
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 255, 1000, 4097):
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert W.base65536_decode(W.base65536_encode(b)) == b
    with pytest.raises(ValueError):
        W.base65536_decode("A")  # U+0041 is not in the alphabet (the reference unwraps a None here: panic)


def test_bf16_rounding_matches_round_to_nearest_even():
    """export/src/main.rs:8-26 is round-half-to-even with quiet NaNs: compare with torch's bfloat16 conversion."""
    import torch

    rng = np.random.default_rng(5)
    x = np.concatenate([
        rng.standard_normal(20000).astype(np.float32) * np.float32(3.0),
        rng.integers(0, 2 ** 32, 20000, dtype=np.uint64).astype(np.uint32).view(np.float32),  # every exponent, NaNs
        np.array([0.0, -0.0, np.inf, -np.inf, 1.0, 1.00390625, 1.01171875, 3.3895314e38, 1e-40], np.float32),
        (np.uint32(0x3F800000) + np.arange(0, 1 << 17, 1 << 15, dtype=np.uint32)).view(np.float32),  # exact ties
    ])
    got = W.f32_to_bf16_bits(x)
    ref = torch.from_numpy(x.copy()).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    nan = np.isnan(x)
    assert np.array_equal(got[~nan], ref[~nan])
    back = W.bf16_bits_to_f32(got)
    assert np.all(np.isnan(back[nan]))
    assert np.array_equal(W.f32_to_bf16_bits(back[~nan]), got[~nan])  # bf16 values survive a second pass


def test_export_text_layout_and_round_trip(golden_dir):
    blob = np.load(os.path.join(golden_dir, "c4net_blob_f32.npy"))
    text = W.export_text(blob, "bf16")
    lines = text.splitlines()
    # serialize_tensors (export/src/main.rs:57-90): loader calls, then the array, two strings per layer in name order
    assert lines[0] == "load_2d(&mut policy.l_1.weight, String::from(PARAMETERS[0]));"
    assert lines[1] == "load_1d(&mut policy.l_1.bias, String::from(PARAMETERS[1]));"
    assert lines[9] == "load_1d(&mut policy.l_5.bias, String::from(PARAMETERS[9]));"
    assert lines[10] == "const PARAMETERS: [&'static str; 10] = ["
    assert lines[11] == "// l_1 - 0" and lines[14] == "// l_2 - 2" and lines[-1] == "];"
    assert len(lines[12]) == 1 + 128 * 63 + 2  # one code point per bf16 value, quoted, trailing comma
    got = W.parse_export_text(text)
    assert np.array_equal(got, W.bf16_bits_to_f32(W.f32_to_bf16_bits(blob)))
    assert np.abs(got - blob).max() < 2.0 ** -8 * np.abs(blob).max()
    # f32 strings (what slimnn::load_1d / load_2d decode: 4 big-endian bytes per value) are lossless
    assert np.array_equal(W.parse_export_text(W.export_text(blob, "f32")), blob)
    w1 = W.split_blob(blob)[0][1]
    s = W.tensor_to_string(w1, "f32")
    raw = W.base65536_decode(s)
    assert len(raw) == 4 * w1.size and np.array_equal(np.frombuffer(raw, ">f4").reshape(w1.shape), w1)
    with pytest.raises(ValueError):
        W.string_to_tensor(s, w1.size + 1)
    with pytest.raises(ValueError):
        W.parse_export_text("const PARAMETERS: [&'static str; 2] = [\n\"\",\n\"\",\n];\n")


def test_ot_varstore_archive_round_trips(golden_dir, tmp_path):
    """`vs.save("models/model_i.ot")` / `vs.load(...)` (synthesis/src/alpha_zero.rs:37,97,194; names policies.rs:20-24): the archive
    libtorch's OutputArchive writes for the committed blob (tests/golden/c4net_blob.ot, made by tests/golden/make_ot_golden.py with the
    image's libtorch — NO reference-produced .ot exists to pin the reader, the reference cannot be built here) reads back bit for bit,
    without torch; and what save_ot writes is an archive torch's own loader (the one InputArchive / vs.load uses) accepts."""
    blob = np.load(os.path.join(golden_dir, "c4net_blob_f32.npy"))
    got = W.load_ot(os.path.join(golden_dir, "c4net_blob.ot"))
    assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), blob.view(np.uint32))
    t = W.load_ot_tensors(os.path.join(golden_dir, "c4net_blob.ot"))
    assert sorted(t) == sorted(f"l_{l}.{k}" for l in range(1, 6) for k in ("weight", "bias")) and t["l_2.weight"].shape == (96, 128)
    # the on-disk names of a reference checkpoint: tch's VarStore::save goes through Tensor::save_multi, which stores `l_1|weight`
    # ('.' -> '|', mapped back by load_multi). The same libtorch writer with those names (c4net_blob_tch_names.ot) reads the same.
    tch = os.path.join(golden_dir, "c4net_blob_tch_names.ot")
    import zipfile

    with zipfile.ZipFile(tch) as z:
        pkl = z.read([n for n in z.namelist() if n.endswith("data.pkl")][0])
    assert b"l_1|weight" in pkl and b"l_1.weight" not in pkl
    got = W.load_ot(tch)
    assert np.array_equal(got.view(np.uint32), blob.view(np.uint32))
    assert sorted(W.load_ot_tensors(tch)) == sorted(t)
    import torch

    for names, sep in (("tch", "|"), ("dotted", ".")):
        mine = str(tmp_path / f"mine_{names}.ot")
        W.save_ot(blob * np.float32(0.5), mine, names=names)
        assert np.array_equal(W.load_ot(mine), blob * np.float32(0.5))
        with zipfile.ZipFile(mine) as z:
            assert f"l_3{sep}bias".encode() in z.read("archive/data.pkl")
        m = torch.jit.load(mine)   # (test-time cross-check only: the product path never imports torch for this)
        sd = dict(m.named_parameters())
        assert np.array_equal(sd[f"l_5{sep}bias"].numpy(), (blob * np.float32(0.5))[-12:]) and tuple(sd[f"l_1{sep}weight"].shape) == (128, 63)
    # a view that reaches past its storage is refused before any memory is touched
    with pytest.raises(ValueError, match="past its storage"):
        W._rebuild_tensor_v2(np.zeros(8, np.float32), 4, (2, 3), (3, 1))
    with pytest.raises(ValueError, match="not a view"):
        W._rebuild_tensor_v2(np.zeros(8, np.float32), -1, (2,), (1,))
    # what is not a Connect4Net VarStore says so
    bad = str(tmp_path / "bad.ot")
    with zipfile.ZipFile(bad, "w") as z:
        z.writestr("x/data.pkl", b"\x80\x02cos\nsystem\nq\x00.")
    with pytest.raises(Exception) as e:
        W.load_ot(bad)
    assert "no business" in str(e.value)
    with pytest.raises(ValueError):
        W.save_ot(blob[:-1], mine)


@pytest.mark.gpu
@pytest.mark.parametrize("container", ["export-bf16", "ot"])
def test_engine_runs_on_an_exported_checkpoint(golden_dir, oracle, container):
    """A PARAMETERS file as the reference's export binary writes it (bf16), or a VarStore archive (.ot, f32: lossless), drives the engine:
    the network outputs equal the oracle's on the loaded weights."""
    import synthesis_amd as sa

    blob = np.load(os.path.join(golden_dir, "c4net_blob_f32.npy"))
    if container == "ot":
        loaded = W.load_ot(os.path.join(golden_dir, "c4net_blob.ot"))
        assert np.array_equal(loaded, blob)
    else:
        loaded = W.parse_export_text(W.export_text(blob, "bf16"))
    eng = sa.Engine(concurrent_games=64, max_explores=64)
    eng.load_weights(loaded)
    from tests.test_gpu_parity import random_positions

    my, op = random_positions(oracle, 40, seed=4, max_moves=30)
    logits, value = eng.policy_eval(my, op)
    fl, fv = oracle.c4net_eval(loaded, my, op, mode=oracle.ACC_FMA)
    assert np.array_equal(logits, fl) and np.array_equal(value, fv)
    eng.close()
