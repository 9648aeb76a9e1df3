"""HIP backend for robot self-collision (reference
``curobo/_src/curobolib/backends/cuda_core_backend/geometry.py:63-227``,
``pybind/geometry_bindings.cpp:16-45``)."""

from __future__ import annotations

import weakref

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .._lib import check, current_stream, load, ptr

#: pair lists at least this long AND at least this dense (enabled / possible pairs) run on the register-tiled
#: bitmap kernel (humanoids); arms keep the LDS pair-list kernels
DENSE_MIN_PAIRS, DENSE_MIN_DENSITY = 8192, 0.25
_bitmap_cache: Dict[Tuple[int, int, int, str], Optional[Tuple[torch.Tensor, int]]] = {}


def pair_bitmap(pair_locations: torch.Tensor, nspheres: int) -> Optional[Tuple[torch.Tensor, int, torch.Tensor, torch.Tensor]]:
    """(bitmap uint32 [2 * nslots, nslots * 64] on the pairs' device, nslots, tile list int32, tile lane masks uint8 [tiles, 64]) for
    ``curobo_hip_self_collision_distance_dense`` -- bit jj of bitmap[jb, i] <-> pair (i, 32 * jb + jj) -- or ``None``
    when the list is not (i, j)-sorted with i < j (then "lowest pair index" is not the lexicographic order the
    dense kernel resolves ties by).  Built once per pair tensor (one host read-back: call outside graph capture)."""
    key = (pair_locations.data_ptr(), int(pair_locations.shape[0]), int(nspheres), str(pair_locations.device))
    hit = _bitmap_cache.get(key)
    if hit is not None and hit[0]() is not None:  # (valid while the tensor it was built from lives: addresses are reused)
        return hit[1]
    p = pair_locations.detach().cpu().numpy().astype(np.int64).reshape(-1, 2)
    i, j = p[:, 0], p[:, 1]
    ok = bool((i < j).all() and (i >= 0).all() and (j < nspheres).all() and (np.diff(i * 65536 + j) > 0).all())
    res = None
    if ok:
        nslots = 4 * ((nspheres + 255) // 256)
        bm = np.zeros((2 * nslots, nslots * 64), np.uint32)
        np.bitwise_or.at(bm, (j // 32, i), (np.uint32(1) << (j % 32).astype(np.uint32)))
        # the 16 x 16 tiles of the pair matrix that hold an enabled pair, (i / 16) | (j / 16) << 8, in (ib, jb) order
        tkey = np.unique((i // 16) * 256 + (j // 16))
        tiles = ((tkey // 256) | ((tkey % 256) << 8)).astype(np.int32)
        # the pair set of every listed tile in the layout in which a lane receives the matrix-core result (self_collision.hip,
        # the mfma kernel): bit reg of masks[c, lane] <-> pair (16 ib + 4 (lane / 16) + reg, 16 jb + lane % 16)
        tix = {int(k): c for c, k in enumerate(tkey)}
        c_of = np.asarray([tix[int(k)] for k in (i // 16) * 256 + (j // 16)], np.int64)
        lane_of = ((i % 16) // 4) * 16 + (j % 16)
        masks = np.zeros((len(tkey), 64), np.uint8)
        np.bitwise_or.at(masks, (c_of, lane_of), (np.uint8(1) << (i % 4).astype(np.uint8)))
        res = (torch.as_tensor(bm.view(np.int32)).to(pair_locations.device).contiguous(), nslots,
               torch.as_tensor(tiles).to(pair_locations.device).contiguous(),
               torch.as_tensor(masks).to(pair_locations.device).contiguous())
    _bitmap_cache[key] = (weakref.ref(pair_locations), res)
    weakref.finalize(pair_locations, _evict_dead, _bitmap_cache, key)  # the device tensors of an entry go with the pair tensor
    return res


def _evict_dead(cache: dict, key) -> None:
    """drop ``cache[key]`` when the tensor it was built from is gone (a live entry under the same key -- the address was
    reused by a new tensor that has been cached since -- stays)"""
    hit = cache.get(key)
    if hit is not None and hit[0]() is None:
        cache.pop(key, None)


def self_collision_distance(
    out_distance: torch.Tensor,
    out_vec: torch.Tensor,
    pair_distance: torch.Tensor,
    sparse_index: torch.Tensor,
    robot_spheres: torch.Tensor,
    sphere_padding: torch.Tensor,
    weight: torch.Tensor,
    pair_locations: torch.Tensor,
    block_batch_max_value: torch.Tensor,
    block_batch_max_index: torch.Tensor,
    num_blocks_per_batch: int,
    max_threads_per_block: int,
    batch_size: int,
    horizon: int,
    nspheres: int,
    num_collision_pairs: int,
    store_pair_distance: bool,
    compute_grad: bool,
):
    """Max sphere-pair penetration per point; modifies the output tensors in place."""
    if (not store_pair_distance and num_collision_pairs >= DENSE_MIN_PAIRS and nspheres <= 1024
            and num_collision_pairs >= DENSE_MIN_DENSITY * 0.5 * nspheres * (nspheres - 1)):
        bm = pair_bitmap(pair_locations, nspheres)
        if bm is not None:
            check(load().curobo_hip_self_collision_distance_dense(
                ptr(out_distance), ptr(out_vec), ptr(sparse_index), ptr(robot_spheres), ptr(sphere_padding), ptr(weight),
                ptr(bm[0]), ptr(bm[2]), int(bm[2].shape[0]), ptr(bm[3]), batch_size, horizon, nspheres, bm[1], int(compute_grad),
                current_stream(out_distance)))
            return
    check(load().curobo_hip_self_collision_distance(
        ptr(out_distance), ptr(out_vec), ptr(pair_distance), ptr(sparse_index), ptr(robot_spheres),
        ptr(sphere_padding), ptr(weight), ptr(pair_locations), ptr(block_batch_max_value),
        ptr(block_batch_max_index), num_blocks_per_batch, max_threads_per_block, batch_size,
        horizon, nspheres, num_collision_pairs, int(store_pair_distance), int(compute_grad),
        current_stream(out_distance),
    ))
