"""Trainer-side sample preparation on the GPU: `get_rotations` and
`get_unique` of the reference (src/tools.py:46-88) for structured sample arrays
(`[('board','f4',(2,S,S)), ('pi','f4',(S*S,)), ('r','f4')]`).

The HIP kernels (csrc/az_augment.hip, reached through the C ABI) do the index
permutations, the key packing and the order-sensitive float32 averaging; the
sort between the two steps of get_unique is torch.sort on the device
(stable, least significant key word first)."""
import numpy as np

from . import engine as _E


def _fields(torch, data, dev):
    n = len(data)
    S = data["board"].shape[-1]
    board = torch.from_numpy(np.ascontiguousarray(data["board"])).to(dev)
    pi = torch.from_numpy(np.ascontiguousarray(data["pi"])).to(dev)
    r = torch.from_numpy(np.ascontiguousarray(data["r"])).to(dev)
    return n, S, board, pi, r


def _check(L, rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc})")


def rotations_device(board, pi, r):
    """Device tensors in, the 7n derived records out (record i*7 + sym)."""
    import torch

    L = _E.load_library()
    n, S = board.shape[0], board.shape[-1]
    size = S * S
    bo = torch.empty((n * 7, 2, S, S), dtype=torch.float32, device=board.device)
    po = torch.empty((n * 7, size), dtype=torch.float32, device=board.device)
    ro = torch.empty(n * 7, dtype=torch.float32, device=board.device)
    stream = torch.cuda.current_stream(board.device).cuda_stream
    _check(L, L.az_aug_rotations(S, n, board.data_ptr(), pi.data_ptr(), r.data_ptr(), bo.data_ptr(), po.data_ptr(),
                                 ro.data_ptr(), stream), "az_aug_rotations")
    return bo, po, ro


def get_rotations(data, device=0):
    """tools.get_rotations (src/tools.py:67-88): data followed by the 7 symmetries of every record."""
    import torch

    dev = torch.device("cuda", device)
    n, S, board, pi, r = _fields(torch, data, dev)
    bo, po, ro = rotations_device(board, pi, r)
    out = np.zeros(n * 8, data.dtype)
    out[:n] = data
    out["board"][n:] = bo.cpu().numpy()
    out["pi"][n:] = po.cpu().numpy()
    out["r"][n:] = ro.cpu().numpy()
    return out


def unique_device(board, pi, r):
    """Device tensors in; one record per distinct board, in np.unique's order."""
    import torch

    L = _E.load_library()
    dev = board.device
    n, S = board.shape[0], board.shape[-1]
    size = S * S
    if n == 0:
        return board, pi, r
    stream = torch.cuda.current_stream(dev).cuda_stream
    words = L.az_aug_key_words(S)
    keys = torch.empty((words, n), dtype=torch.int64, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    _check(L, L.az_aug_pack_keys(S, n, board.data_ptr(), keys.data_ptr(), bad.data_ptr(), stream), "az_aug_pack_keys")
    if int(bad.item()):
        raise ValueError("get_unique on the GPU needs boards made of 0 and 1")
    keys ^= -(1 << 63)  # unsigned order through the signed sort
    perm = torch.arange(n, dtype=torch.int64, device=dev)
    for w in range(words - 1, -1, -1):  # stable sorts, least significant word first
        _, idx = torch.sort(keys[w][perm], stable=True)
        perm = perm[idx]
    sorted_keys = keys[:, perm]
    new_seg = torch.ones(n, dtype=torch.bool, device=dev)
    if n > 1:
        new_seg[1:] = (sorted_keys[:, 1:] != sorted_keys[:, :-1]).any(dim=0)
    starts = torch.nonzero(new_seg).flatten()
    n_unique = int(starts.numel())
    seg_start = torch.cat([starts, torch.tensor([n], dtype=torch.int64, device=dev)]).to(torch.int32)
    members = perm.to(torch.int32)
    bo = torch.empty((n_unique, 2, S, S), dtype=torch.float32, device=dev)
    po = torch.empty((n_unique, size), dtype=torch.float32, device=dev)
    ro = torch.empty(n_unique, dtype=torch.float32, device=dev)
    _check(L, L.az_aug_unique_average(S, n_unique, members.data_ptr(), seg_start.data_ptr(), board.data_ptr(),
                                      pi.data_ptr(), r.data_ptr(), bo.data_ptr(), po.data_ptr(), ro.data_ptr(),
                                      stream), "az_aug_unique_average")
    return bo, po, ro


def get_unique(data, device=0):
    """tools.get_unique (src/tools.py:46-65): duplicate boards merged, pi and r averaged."""
    import torch

    dev = torch.device("cuda", device)
    n, S, board, pi, r = _fields(torch, data, dev)
    bo, po, ro = unique_device(board, pi, r)
    out = np.zeros(bo.shape[0], data.dtype)
    out["board"] = bo.cpu().numpy()
    out["pi"] = po.cpu().numpy()
    out["r"] = ro.cpu().numpy()
    return out
