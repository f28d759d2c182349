"""Image preprocessor (cm_image_preprocess, host only): the reference's own KATs (qwen3_5/processor.rs:262-329), the numpy
oracle, and -- for the resampling the reference says it mirrors -- PIL's BICUBIC bit for bit."""
import numpy as np
import pytest

from crane_amd import _lib
from crane_amd.processor import PreprocessorConfig, batch_images
from oracle import preprocess_oracle as PO


def kat_cfg():
    """processor.rs:266-279: min_pixels tiny so the small synthetic images are not resized"""
    return PreprocessorConfig(shortest_edge=16, longest_edge=16777216, patch_size=16, temporal_patch_size=2, merge_size=2,
                              image_mean=[0.5, 0.5, 0.5], image_std=[0.5, 0.5, 0.5])


def test_smart_resize_rounds_to_nearest_not_up():
    """processor.rs:284-297, verbatim"""
    pc = PreprocessorConfig(shortest_edge=65536, longest_edge=16777216)
    assert pc.smart_resize(294, 1024) == (288, 1024)
    assert pc.smart_resize(300, 1024) == (288, 1024)
    assert pc.smart_resize(310, 1024) == (320, 1024)
    assert pc.smart_resize(288, 1024) == (288, 1024)
    for h, w in [(294, 1024), (37, 5000), (9000, 9000), (10, 10), (448, 448), (1, 70000)]:
        assert pc.smart_resize(h, w) == PO.smart_resize(h, w, 32, 65536, 16777216), (h, w)


def test_patch_layout_is_merge_block_major_and_channel_major():
    """processor.rs:305-329, verbatim: 4 x 2 patches, the raster patch index encoded in the red channel"""
    pc = kat_cfg()
    P, wp, hp = 16, 4, 2
    img = np.zeros((hp * P, wp * P, 3), np.uint8)
    for y in range(hp * P):
        for x in range(wp * P):
            img[y, x] = ((y // P) * wp + x // P, 7, 9)
    pix, grid = pc.process(img)
    assert grid == (1, hp, wp) and pix.shape == (hp * wp, 3 * 2 * P * P)
    dec = lambda v: int(round((v * 0.5 + 0.5) * 255.0))
    assert [dec(r[0]) for r in pix] == [0, 1, 4, 5, 2, 3, 6, 7]              # merge-block-major
    pp = P * P
    for r in pix:
        for c, want in ((1, 7), (2, 9)):
            for t in range(2):
                assert dec(r[c * 2 * pp + t * pp]) == want                   # (channel, temporal, y, x)


@pytest.mark.parametrize("hw", [(448, 448), (300, 500), (97, 1031), (640, 480), (33, 33)])
def test_matches_oracle_and_pil_resize(hw):
    rng = np.random.default_rng(hw[0] * 131 + hw[1])
    img = rng.integers(0, 256, size=(hw[0], hw[1], 3), dtype=np.uint8)
    img[: hw[0] // 2] = np.clip(img[: hw[0] // 2].astype(np.int32) // 4 + np.arange(hw[1])[None, :, None] % 200, 0, 255).astype(np.uint8)
    pc = PreprocessorConfig(shortest_edge=65536, longest_edge=1048576,
                            image_mean=[0.48145466, 0.4578275, 0.40821073], image_std=[0.26862954, 0.26130258, 0.27577711])
    pix, grid = pc.process(img)
    ref, rgrid = PO.process(img, min_pixels=65536, max_pixels=1048576, mean=pc.image_mean, std=pc.image_std)
    assert grid == rgrid and pix.shape == ref.shape
    # the resize is PIL's fixed-point arithmetic restated: identical bytes, so identical floats
    assert np.array_equal(pix, ref), float(np.abs(pix - ref).max())


def test_errors_and_batching():
    pc = kat_cfg()
    with pytest.raises(ValueError):
        pc.process(np.zeros((32, 32), np.uint8))
    bad = PreprocessorConfig(image_std=[0.5, 0.0, 0.5])
    with pytest.raises(_lib.CraneError):
        bad.process(np.zeros((32, 32, 3), np.uint8))
    a = pc.process(np.zeros((32, 64, 3), np.uint8))
    b = pc.process(np.full((64, 32, 3), 255, np.uint8))
    pix, grid = batch_images([a, b])
    assert pix.shape == (16, 1536) and grid.tolist() == [[1, 2, 4], [1, 4, 2]]
    assert np.all(pix[:8] == -1.0) and np.all(pix[8:] == 1.0)


def test_hostile_configurations_are_answered_with_a_status_code():
    """preprocessor_config.json comes from the model directory: whatever it says, cm_image_preprocess / cm_image_smart_resize
    return a status (no out-of-range float -> int cast, no allocation request the size of min_pixels, no exception across the
    C ABI).  Run in a subprocess so that an abort would fail this test rather than pytest."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import numpy as np
        from crane_amd import _lib
        from crane_amd.processor import PreprocessorConfig
        img = np.zeros((48, 80, 3), np.uint8)
        answered = 0
        for kw in [dict(shortest_edge=10**18), dict(shortest_edge=2**63), dict(shortest_edge=2**40, longest_edge=2**41),
                   dict(longest_edge=1), dict(patch_size=2**31), dict(patch_size=65536, merge_size=65536), dict(merge_size=2**20),
                   dict(temporal_patch_size=2**31), dict(patch_size=1024, merge_size=64, shortest_edge=2**50),
                   dict(shortest_edge=3 * 10**8, longest_edge=4 * 10**8)]:
            pc = PreprocessorConfig(**kw)
            for call in (lambda: pc.smart_resize(48, 80), lambda: pc.process(img)):
                try:
                    call()
                except _lib.CraneError:
                    pass
                answered += 1
        ok = PreprocessorConfig(shortest_edge=16).process(img)          # and a sane one still works afterwards
        assert ok[0].shape == (24, 1536) and ok[1] == (1, 4, 6), (ok[0].shape, ok[1])      # 48 x 80 -> 64 x 96 (nearest multiples of 32)
        print("answered", answered)
    """)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=str(__import__("pathlib").Path(__file__).parent.parent))
    assert p.returncode == 0 and "answered 20" in p.stdout, (p.returncode, p.stdout[-1000:], p.stderr[-2000:])


def test_random_configurations_under_address_sanitizer(tmp_path):
    """tests/fuzz_preprocess.cpp: 1500 random (configuration, image) pairs through cm_image_preprocess compiled with
    -fsanitize=address,undefined, the output buffer sized exactly -- the resampling loops index three buffers by hand."""
    import os, shutil, subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "ppfuzz")
    b = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        "-I", os.path.join(root, "include"), os.path.join(root, "tests", "fuzz_preprocess.cpp"),
                        os.path.join(root, "crane_amd", "csrc", "image_preprocess.cpp"), "-o", exe],
                       capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and "sanitize" in b.stderr.lower():
        pytest.skip("g++ without the sanitizer runtimes")
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe, "5", "1500"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "1500 ok" in r.stdout, (r.returncode, r.stdout[-300:], r.stderr[-3000:])
