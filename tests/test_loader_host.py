"""Host logic of the loaders, no GPU: shard discovery + header parsing of safetensors checkpoints (cm_checkpoint_inspect) and the config the C++ side derives from a file's metadata and tensor directory
(cm_gguf_config; qwen3/model.rs:138-147, qwen3_5/model.rs:196-287) and the reader's error behaviour.  Files are written by
oracle/gguf_oracle.py from the synthetic configs."""
import struct

import numpy as np
import pytest

from crane_amd import _lib, configs, synth
from crane_amd.backend import gguf_config
from oracle import gguf_oracle as G

DENSE_KEYS = ("hidden_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads", "head_dim",
              "intermediate_size", "vocab_size", "max_position_embeddings", "tie_word_embeddings")
HYBRID_KEYS = DENSE_KEYS + ("full_attention_interval", "linear_conv_kernel_dim", "linear_key_head_dim", "linear_value_head_dim",
                            "linear_num_key_heads", "linear_num_value_heads", "attn_output_gate")


@pytest.mark.parametrize("name", ["tiny-qwen3", "tiny-qwen3-untied"])
@pytest.mark.parametrize("kind", ["q8_0", "q4_k"])
def test_dense_config_round_trip(tmp_path, name, kind):
    cfg = configs.get_config(name)
    w = synth.synth_weights_f32(cfg, seed=0)
    path = str(tmp_path / "m.gguf")
    G.write_qwen3_gguf(path, cfg, w, lambda n, s: G.TYPE_NAMES[kind])
    got = gguf_config(path)
    assert got["model_type"] == "qwen3"
    for k in DENSE_KEYS:
        assert got[k] == cfg[k], (k, got[k], cfg[k])       # tied head: detected from the missing output.weight
    assert abs(got["rope_theta"] - cfg["rope_theta"]) < 1 and abs(got["rms_norm_eps"] - cfg["rms_norm_eps"]) < 1e-9


def test_hybrid_config_round_trip(tmp_path):
    cfg = configs.get_config("tiny-qwen3.5")
    w = synth.synth_weights_f32(cfg, seed=0)
    path = str(tmp_path / "h.gguf")
    G.write_qwen35_gguf(path, cfg, w, lambda n, s: G.GGML_Q8_0)
    got = gguf_config(path)
    assert got["model_type"] == "qwen3_5_text"
    for k in HYBRID_KEYS:
        assert got[k] == cfg[k], (k, got[k], cfg[k])       # layer kinds from blk.i.ssm_a, output gate from attn_q's rows
    rp, want = got["rope_parameters"], cfg["rope_parameters"]
    assert rp["mrope_section"] == want["mrope_section"] and rp["partial_rotary_factor"] == want["partial_rotary_factor"]
    assert abs(rp["rope_theta"] - want["rope_theta"]) < 1


def _minimal(path, arch="qwen3", version=3, magic=0x46554747, drop=None, extra_tensor=None):
    """A header-only GGUF (metadata + one tiny tensor) for the error paths."""
    md = G.qwen3_metadata(configs.get_config("tiny-qwen3"))
    md["general.architecture"] = (G.T_STR, arch)
    if drop:
        md.pop(drop)
    tensors = [("token_embd.weight", np.zeros((512, 256), np.float32), G.GGML_F32)]
    if extra_tensor:
        tensors.append(extra_tensor)
    G.write_gguf(path, md, tensors)
    if magic != 0x46554747 or version != 3:
        with open(path, "r+b") as f:
            f.write(struct.pack("<II", magic, version))


def test_reader_errors(tmp_path):
    p = str(tmp_path / "x.gguf")
    with pytest.raises(_lib.CraneError):
        gguf_config(str(tmp_path / "missing.gguf"))
    _minimal(p, magic=0x12345678)
    with pytest.raises(_lib.CraneError, match="not a GGUF"):
        gguf_config(p)
    _minimal(p, version=1)
    with pytest.raises(_lib.CraneError, match="version"):
        gguf_config(p)
    _minimal(p, arch="llama")
    with pytest.raises(_lib.CraneError, match="llama"):
        gguf_config(p)
    _minimal(p, drop="qwen3.block_count")
    with pytest.raises(_lib.CraneError, match="block_count"):
        gguf_config(p)
    _minimal(p)
    with open(p, "r+b") as f:                                  # cut the file inside the tensor data
        f.truncate(200)
    with pytest.raises(_lib.CraneError):
        gguf_config(p)


def test_buffer_protocol():
    import ctypes as C
    lib = _lib.load()
    need = C.c_size_t(0)
    assert lib.cm_gguf_config(None, None, 0, C.byref(need)) != 0          # null path


def _fnv1a(b: bytes) -> int:
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.parametrize("shards", [1, 3])
def test_safetensors_shard_discovery(tmp_path, shards):
    """cm_checkpoint_inspect: index.json weight_map / single file / header parsing (utils/utils.rs:16-57) in the C++ loader,
    checked tensor by tensor (dtype, shape, byte hash) against the files read back in Python."""
    import json, os
    from crane_amd.backend import checkpoint_inspect
    cfg = configs.get_config("tiny-qwen3-untied")
    d = synth.write_model_dir(str(tmp_path / "m"), cfg, seed=0, shards=shards)
    got = checkpoint_inspect(d)
    want = {}
    for fn in sorted(os.listdir(d)):
        if not fn.endswith(".safetensors"):
            continue
        raw = open(os.path.join(d, fn), "rb").read()
        (hl,) = struct.unpack("<Q", raw[:8])
        hdr = json.loads(raw[8:8 + hl])
        for name, t in hdr.items():
            if name == "__metadata__":
                continue
            b, e = t["data_offsets"]
            want[name] = (t["dtype"], t["shape"], _fnv1a(raw[8 + hl + b: 8 + hl + e]) if e - b <= 1 << 16 else None, e - b)
    assert set(got) == set(want) == {n for n, *_ in synth.specs_for(cfg)}
    for name, (dt, shape, h, nb) in want.items():
        g = got[name]
        assert g["dtype"] == dt and g["shape"] == shape and g["nbytes"] == nb
        if h is not None:
            assert int(g["fnv1a"]) == h, name


def test_safetensors_errors(tmp_path):
    from crane_amd.backend import checkpoint_inspect
    with pytest.raises(_lib.CraneError):
        checkpoint_inspect(str(tmp_path / "nope"))
    d = tmp_path / "empty"
    d.mkdir()
    with pytest.raises(_lib.CraneError, match="no safetensors"):
        checkpoint_inspect(str(d))
    bad = tmp_path / "bad"
    bad.mkdir()
    (bad / "model.safetensors").write_bytes(struct.pack("<Q", 1 << 40) + b"{}")
    with pytest.raises(_lib.CraneError, match="header length"):
        checkpoint_inspect(str(bad))
    (bad / "model.safetensors").write_bytes(b"\x01")
    with pytest.raises(_lib.CraneError, match="truncated"):
        checkpoint_inspect(str(bad))
    hdr = b'{"w": {"dtype": "F32", "shape": [4], "data_offsets": [0, 64]}}'
    (bad / "model.safetensors").write_bytes(struct.pack("<Q", len(hdr)) + hdr + b"\0" * 16)
    with pytest.raises(_lib.CraneError, match="out of file bounds"):
        checkpoint_inspect(str(bad))


def test_safetensors_hostile_headers(tmp_path):
    """The checkpoint parser reads an mmap'ed, attacker-controlled file: every length is checked without overflow, every
    tensor's byte range must equal shape x dtype, and JSON nesting is bounded."""
    from crane_amd.backend import checkpoint_inspect
    bad = tmp_path / "hostile"
    bad.mkdir()
    f = bad / "model.safetensors"

    def put(hdr: bytes, data: bytes = b"\0" * 64, hl=None):
        f.write_bytes(struct.pack("<Q", len(hdr) if hl is None else hl) + hdr + data)

    put(b"{}", hl=(1 << 64) - 4)                                 # 8 + hl wraps around
    with pytest.raises(_lib.CraneError, match="header length"):
        checkpoint_inspect(str(bad))
    put(b'{"w": {"dtype": "F32", "shape": [4, 8], "data_offsets": [0, 64]}}')       # 32 elements x 4 B != 64 B
    with pytest.raises(_lib.CraneError, match="do not match"):
        checkpoint_inspect(str(bad))
    put(b'{"w": {"dtype": "F16", "shape": [4294967296, 4294967296, 4], "data_offsets": [0, 64]}}')    # numel overflows int64
    with pytest.raises(_lib.CraneError, match="bad shape|do not match"):
        checkpoint_inspect(str(bad))
    put(b'{"w": {"dtype": "BF16", "shape": [-4], "data_offsets": [0, 8]}}')
    with pytest.raises(_lib.CraneError, match="bad dimension"):
        checkpoint_inspect(str(bad))
    put(b'{"w": {"dtype": "BF16", "shape": [4], "data_offsets": [8, 0]}}')
    with pytest.raises(_lib.CraneError, match="out of file bounds"):
        checkpoint_inspect(str(bad))
    deep = b"[" * 5000 + b"]" * 5000
    put(b'{"__metadata__": ' + deep + b"}")
    with pytest.raises(_lib.CraneError, match="nesting too deep"):
        checkpoint_inspect(str(bad))
    put(b'{"w": {"dtype": "F32", "shape": [4, 4], "data_offsets": [0, 64]}}')       # and a well-formed one still loads
    assert checkpoint_inspect(str(bad))["w"]["nbytes"] == 64


def test_gguf_hostile_tensor_directory(tmp_path):
    """Tensor shapes / offsets in a GGUF directory cannot overflow the bounds check."""
    p = str(tmp_path / "x.gguf")
    _minimal(p)
    raw = bytearray(open(p, "rb").read())
    # the directory entry of token_embd.weight: name, n_dims = 2, dims [256, 512] (innermost first), type, offset
    at = raw.find(b"token_embd.weight") + len(b"token_embd.weight")
    assert struct.unpack_from("<I", raw, at)[0] == 2
    struct.pack_into("<QQ", raw, at + 4, 1 << 33, 1 << 33)      # 2^66 elements: the product overflows uint64
    open(p, "wb").write(bytes(raw))
    with pytest.raises(_lib.CraneError, match="overflows|past the end"):
        gguf_config(p)
    _minimal(p)
    raw = bytearray(open(p, "rb").read())
    struct.pack_into("<Q", raw, at + 4 + 16 + 4, (1 << 64) - 64)   # offset near 2^64: data0 + off would wrap
    open(p, "wb").write(bytes(raw))
    with pytest.raises(_lib.CraneError, match="past the end"):
        gguf_config(p)


def test_fuzzed_files_through_the_abi(tmp_path):
    """3000 mutated GGUF / safetensors files (tests/fuzz_host_parsers.py: bit flips, boundary values in length / offset fields,
    truncations, structured edits of the safetensors header) through cm_gguf_config / cm_checkpoint_inspect in a SUBPROCESS:
    every file is answered with a status code; a crash or hang of the parser fails here instead of killing pytest."""
    import os, subprocess, sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_host_parsers.py")
    p = subprocess.run([sys.executable, script, "1", "3000", str(tmp_path / "w")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    assert "3000 files" in p.stdout


def test_fuzzed_files_under_address_sanitizer(tmp_path):
    """The same mutations against the two container parsers compiled with -fsanitize=address,undefined
    (tests/fuzz_harness.cpp: mmap replaced by an exact-size heap copy, so one byte past the end of the file is a report)."""
    import os, shutil, subprocess, sys
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / "fuzz_harness")
    b = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        "-I", os.path.join(here, "..", "crane_amd", "csrc"), os.path.join(here, "fuzz_harness.cpp"), "-o", exe],
                       capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and ("asan" in b.stderr.lower() or "sanitize" in b.stderr.lower()):
        pytest.skip("g++ without the sanitizer runtimes")
    assert b.returncode == 0, b.stderr[-3000:]
    indir = str(tmp_path / "in")
    e = subprocess.run([sys.executable, os.path.join(here, "fuzz_host_parsers.py"), "2", "3000", indir, "--emit"],
                       capture_output=True, text=True, timeout=300)
    assert e.returncode == 0, e.stderr[-2000:]
    r = subprocess.run([exe, indir], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "3000 inputs, no sanitizer report" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
