"""Host-side mirror of the reference's image preprocessor on top of the C ABI
(``PreprocessorConfig`` / ``process`` / ``batch_images``, crane-core/src/models/qwen3_5/processor.rs:20-235).
All arithmetic happens in ``libcrane_mi355.so`` (cm_image_preprocess); this module only marshals."""
from __future__ import annotations

import ctypes as C
import json
from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

import numpy as np

from . import _lib


@dataclass
class PreprocessorConfig:
    """preprocessor_config.json (processor.rs:20-35); size.shortest_edge / longest_edge are pixel COUNTS."""
    shortest_edge: int = 65536
    longest_edge: int = 16777216
    patch_size: int = 16
    temporal_patch_size: int = 2
    merge_size: int = 2
    image_mean: Sequence[float] = field(default_factory=lambda: [0.5, 0.5, 0.5])
    image_std: Sequence[float] = field(default_factory=lambda: [0.5, 0.5, 0.5])

    @classmethod
    def load(cls, model_dir: str) -> "PreprocessorConfig":
        """load_preprocessor_config (processor.rs:44-52)"""
        with open(f"{model_dir}/preprocessor_config.json") as f:
            j = json.load(f)
        return cls(shortest_edge=j["size"]["shortest_edge"], longest_edge=j["size"]["longest_edge"], patch_size=j["patch_size"],
                   temporal_patch_size=j["temporal_patch_size"], merge_size=j["merge_size"], image_mean=j["image_mean"],
                   image_std=j["image_std"])

    def factor(self) -> int:
        return self.patch_size * self.merge_size

    def _c(self):
        c = _lib.CmPreprocConfig()
        c.patch_size, c.temporal_patch_size, c.merge_size = self.patch_size, self.temporal_patch_size, self.merge_size
        c.min_pixels, c.max_pixels = self.shortest_edge, self.longest_edge
        for i in range(3):
            c.image_mean[i] = self.image_mean[i]
            c.image_std[i] = self.image_std[i]
        return c

    def smart_resize(self, height: int, width: int) -> Tuple[int, int]:
        lib = _lib.load()
        h, w = C.c_uint32(), C.c_uint32()
        c = self._c()
        rc = lib.cm_image_smart_resize(C.byref(c), height, width, C.byref(h), C.byref(w))
        if rc != 0:
            raise _lib.CraneError(rc, lib.cm_preprocess_last_error().decode())
        return int(h.value), int(w.value)

    def process(self, image_rgb: np.ndarray):
        """process (processor.rs:114-210): uint8 [H, W, 3] -> (pixel_values [n_patches, 3 * T * P * P] f32, (t, h, w))."""
        lib = _lib.load()
        img = np.ascontiguousarray(image_rgb, dtype=np.uint8)
        if img.ndim != 3 or img.shape[2] != 3:
            raise ValueError("expected an RGB image [H, W, 3]")
        c = self._c()
        grid = (C.c_uint32 * 3)()
        n = C.c_size_t()
        rc = lib.cm_image_preprocess(C.byref(c), img.ctypes.data_as(C.c_char_p), img.shape[0], img.shape[1], None, 0, grid, C.byref(n))
        if rc != 0:
            raise _lib.CraneError(rc, lib.cm_preprocess_last_error().decode())
        in_dim = 3 * self.temporal_patch_size * self.patch_size * self.patch_size
        out = np.empty((n.value, in_dim), dtype=np.float32)
        rc = lib.cm_image_preprocess(C.byref(c), img.ctypes.data_as(C.c_char_p), img.shape[0], img.shape[1],
                                     out.ctypes.data_as(C.POINTER(C.c_float)), out.size, grid, C.byref(n))
        if rc != 0:
            raise _lib.CraneError(rc, lib.cm_preprocess_last_error().decode())
        return out, (int(grid[0]), int(grid[1]), int(grid[2]))


def batch_images(images: List[Tuple[np.ndarray, Tuple[int, int, int]]]):
    """batch_images (processor.rs:214-235): flat pixel_values [sum patches, in_dim] + grid_thw [n, 3]."""
    pix = np.concatenate([p for p, _ in images], axis=0) if images else np.zeros((0, 0), np.float32)
    grid = np.asarray([g for _, g in images], dtype=np.uint32).reshape(-1, 3)
    return pix, grid
