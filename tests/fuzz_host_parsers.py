"""Mutation fuzzer for the two host-only parsers behind the C ABI (no GPU): cm_gguf_config (GGUF v3 container: metadata
key/values + tensor directory, csrc/gguf.h) and cm_checkpoint_inspect (safetensors header JSON + index.json, csrc/safetensors.h,
csrc/json_min.h).  Both read an mmap'ed, caller-supplied file; whatever the bytes are, the call must come back with a status
code -- never crash, hang or read out of bounds.

    python tests/fuzz_host_parsers.py <seed> <iterations> [workdir]            mutate + parse through the C ABI
    python tests/fuzz_host_parsers.py <seed> <iterations> <outdir> --emit      only write the mutated files
                                                                               (f%05d.gguf, s%05d/model.safetensors) for
                                                                               the AddressSanitizer harness, tests/fuzz_harness.cpp

Run as a SUBPROCESS by tests/test_loader_host.py (a crash of the parser must fail a test, not kill pytest).  Exit code 0 =
every mutated file was answered with success or a CraneError; the last mutation is kept in <workdir> when something else
happens, and the process dies with the signal if the parser crashes."""
import json
import os
import struct
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crane_amd import _lib, configs                                    # noqa: E402
from crane_amd.backend import checkpoint_inspect, gguf_config           # noqa: E402
from oracle import gguf_oracle as G                                     # noqa: E402

INTERESTING = [0, 1, 2, 3, 4, 7, 8, 16, 31, 32, 33, 255, 256, 1 << 15, 1 << 16, (1 << 31) - 1, 1 << 31, (1 << 32) - 1, 1 << 32,
               1 << 40, (1 << 63) - 1, 1 << 63, (1 << 64) - 64, (1 << 64) - 8, (1 << 64) - 1]


def mutate(raw: bytearray, limit: int, r: np.random.Generator) -> bytearray:
    """One to four edits inside the first `limit` bytes (the structured part of the file)."""
    out = bytearray(raw)
    for _ in range(int(r.integers(1, 5))):
        kind = int(r.integers(0, 6))
        at = int(r.integers(0, max(1, min(limit, len(out)))))
        if kind == 0:                                                   # bit flip
            out[at] ^= 1 << int(r.integers(0, 8))
        elif kind == 1:                                                 # random byte
            out[at] = int(r.integers(0, 256))
        elif kind == 2 and at + 8 <= len(out):                          # 64-bit field <- boundary value
            struct.pack_into("<Q", out, at, INTERESTING[int(r.integers(0, len(INTERESTING)))])
        elif kind == 3 and at + 4 <= len(out):                          # 32-bit field <- boundary value
            struct.pack_into("<I", out, at, INTERESTING[int(r.integers(0, len(INTERESTING)))] & 0xFFFFFFFF)
        elif kind == 4:                                                 # truncation
            out = out[:at]
            if not out:
                out = bytearray(b"\0")
        else:                                                           # duplicate a span over another
            n = int(r.integers(1, 64))
            src = int(r.integers(0, max(1, min(limit, len(out)))))
            out[at:at + n] = out[src:src + n]
    return out


def structured_json_mutation(hdr: dict, r: np.random.Generator) -> bytes:
    """Edits on the parsed safetensors header (byte flips rarely leave the JSON well-formed enough to reach the checks)."""
    h = json.loads(json.dumps(hdr))
    names = [k for k in h if k != "__metadata__"]
    for _ in range(int(r.integers(1, 4))):
        t = h[names[int(r.integers(0, len(names)))]]
        what = int(r.integers(0, 6))
        v = INTERESTING[int(r.integers(0, len(INTERESTING)))]
        if what == 0:
            t["shape"] = [v if r.random() < 0.5 else -v for _ in range(int(r.integers(0, 5)))]
        elif what == 1:
            t["data_offsets"] = [v, INTERESTING[int(r.integers(0, len(INTERESTING)))]]
        elif what == 2:
            t["dtype"] = ["F32", "F16", "BF16", "I8", "U8", "F64", "I64", "BOOL", "", "Q4_K", 7, None][int(r.integers(0, 12))]
        elif what == 3:
            t["shape"] = "x" if r.random() < 0.5 else {"a": [1]}
        elif what == 4:
            t.pop(["dtype", "shape", "data_offsets"][int(r.integers(0, 3))], None)
        else:
            h[names[0] * int(r.integers(1, 300))] = t
    s = json.dumps(h)
    if r.random() < 0.2:
        s = s.replace("[", "[" * int(r.integers(1, 3000)), 1)
    if r.random() < 0.2:
        s = s[: int(r.integers(0, len(s) + 1))]
    return s.encode()


def main():
    seed, iters = int(sys.argv[1]), int(sys.argv[2])
    work = sys.argv[3] if len(sys.argv) > 3 else tempfile.mkdtemp(prefix="cmfuzz")
    emit = "--emit" in sys.argv[4:]
    os.makedirs(work, exist_ok=True)
    r = np.random.default_rng(seed)

    # ---- seeds: a header-only GGUF (qwen3 metadata + one F32 tensor) and a two-tensor safetensors file ----
    gpath = os.path.join(work, "f.gguf")
    md = G.qwen3_metadata(configs.get_config("tiny-qwen3"))
    G.write_gguf(gpath, md, [("token_embd.weight", np.zeros((512, 256), np.float32), G.GGML_F32),
                             ("blk.0.attn_q.weight", np.zeros((4, 32), np.float32), G.GGML_F32)])
    graw = bytearray(open(gpath, "rb").read())
    gguf_config(gpath)                                                   # the seed itself parses
    glimit = len(graw) - 512 * 256 * 4 - 4 * 32 * 4                      # metadata + directory (+ alignment padding)
    if emit:                                                             # a small seed: hundreds of files are written
        G.write_gguf(gpath, md, [("token_embd.weight", np.zeros((8, 32), np.float32), G.GGML_F32),
                                 ("blk.0.attn_q.weight", np.zeros((4, 32), np.float32), G.GGML_F32)])
        graw = bytearray(open(gpath, "rb").read())
        glimit = len(graw) - 8 * 32 * 4 - 4 * 32 * 4
        os.remove(gpath)
    sdir = os.path.join(work, "st")
    os.makedirs(sdir, exist_ok=True)
    spath = os.path.join(sdir, "model.safetensors")
    hdr = {"__metadata__": {"format": "pt"},
           "a.weight": {"dtype": "F32", "shape": [4, 4], "data_offsets": [0, 64]},
           "b.weight": {"dtype": "BF16", "shape": [8], "data_offsets": [64, 80]}}
    hb = json.dumps(hdr).encode()
    sraw = bytearray(struct.pack("<Q", len(hb)) + hb + b"\0" * 80)
    open(spath, "wb").write(bytes(sraw))
    assert checkpoint_inspect(sdir)["a.weight"]["nbytes"] == 64

    ok = err = 0
    if emit:
        os.remove(spath); os.rmdir(sdir)
        for it in range(iters):
            which = it % 3
            if which == 0:
                open(os.path.join(work, f"f{it:05d}.gguf"), "wb").write(bytes(mutate(graw, glimit, r)))
                continue
            d = os.path.join(work, f"s{it:05d}")
            os.makedirs(d, exist_ok=True)
            if which == 1:
                open(os.path.join(d, "model.safetensors"), "wb").write(bytes(mutate(sraw, 8 + len(hb), r)))
            else:
                body = structured_json_mutation(hdr, r)
                hl = len(body) if r.random() < 0.8 else INTERESTING[int(r.integers(0, len(INTERESTING)))]
                open(os.path.join(d, "model.safetensors"), "wb").write(struct.pack("<Q", hl) + body + b"\0" * int(r.integers(0, 96)))
        print(f"fuzz seed {seed}: {iters} files written to {work}")
        return
    for it in range(iters):
        which = it % 3
        try:
            if which == 0:
                open(gpath, "wb").write(bytes(mutate(graw, glimit, r)))
                gguf_config(gpath)
            elif which == 1:
                open(spath, "wb").write(bytes(mutate(sraw, 8 + len(hb), r)))
                checkpoint_inspect(sdir)
            else:
                body = structured_json_mutation(hdr, r)
                hl = len(body) if r.random() < 0.8 else INTERESTING[int(r.integers(0, len(INTERESTING)))]
                open(spath, "wb").write(struct.pack("<Q", hl) + body + b"\0" * int(r.integers(0, 96)))
                checkpoint_inspect(sdir)
            ok += 1
        except _lib.CraneError:
            err += 1
        except (UnicodeDecodeError, json.JSONDecodeError):
            err += 1                                                    # a mutated name that is no longer UTF-8 / JSON in the REPLY
    print(f"fuzz seed {seed}: {iters} files, {ok} parsed, {err} rejected")


if __name__ == "__main__":
    main()
