"""CPU: the N>1 path (slice ownership + result gather) with world_size 2 and 3 over gloo.
The per-item compute is stood in by the C oracle so that the gathered statuses can be checked
against a single-process run item by item."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ark_ec_vrfs_amd.sharding import gather_results, shard_range


def test_shard_ranges_tile_exactly():
    for n in (0, 1, 7, 64, 1000, (1 << 20) + 3):
        for world in (1, 2, 3, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            for (a, b), (c, d) in zip(edges, edges[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import c_oracle as co
        d = np.load(os.path.join(tmp, "batch.npz"))
        lo, hi = shard_range(n, rank, world)
        st = co.ietf_verify_batch(d["pk"][lo:hi], d["h"][lo:hi], d["g"][lo:hi], d["c"][lo:hi], d["s"][lo:hi], b"")
        full = gather_results(torch.from_numpy(st), n, rank, world)
        dist.barrier()
        np.save(os.path.join(tmp, f"out{rank}.npy"), full.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_verify_equals_single_process(tmp_path, synth, world):
    from oracle import c_oracle as co
    n = 13
    sk, msg = synth(n, start=300)
    r = co.ietf_prove_batch(sk, msgs=msg, ad=b"", threads=4)
    s = r["s"].copy(); s[[2, 7, 12], 0] ^= 1                      # three bad proofs
    np.savez(os.path.join(tmp_path, "batch.npz"), pk=r["pk"], h=r["input"], g=r["output"], c=r["c"], s=s)
    ref = co.ietf_verify_batch(r["pk"], r["input"], r["output"], r["c"], s, b"", threads=4)
    assert list(np.nonzero(ref)[0]) == [2, 7, 12]
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        got = np.load(os.path.join(tmp_path, f"out{rank}.npy"))
        assert got.shape == (n,) and (got == ref).all()


def _worker_p256(rank, world, port, n, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import c_oracle as co
        d = np.load(os.path.join(tmp, "batch.npz"))
        lo, hi = shard_range(n, rank, world)
        r = co.p256_ietf_prove_batch(d["sk"][lo:hi], msgs=d["msg"][lo:hi], ad=b"shard")
        rows = np.concatenate([r["output"], r["c"], r["s"]], axis=1)          # 33 + 32 + 32 bytes per item
        full = gather_results(torch.from_numpy(rows), n, rank, world)
        dist.barrier()
        np.save(os.path.join(tmp, f"out{rank}.npy"), full.numpy())
    finally:
        dist.destroy_process_group()


def test_sharded_secp256r1_prove_gathers_sec1_rows(tmp_path):
    """The gather is width-agnostic: 97-byte rows (a 33-byte Sec1 output + two big-endian scalars) from ragged shards."""
    from oracle import c_oracle as co
    n, world = 11, 2
    rng = np.random.default_rng(4)
    sk = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    msg = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    np.savez(os.path.join(tmp_path, "batch.npz"), sk=sk, msg=msg)
    r = co.p256_ietf_prove_batch(sk, msgs=msg, ad=b"shard", threads=2)
    ref = np.concatenate([r["output"], r["c"], r["s"]], axis=1)
    mp.spawn(_worker_p256, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        got = np.load(os.path.join(tmp_path, f"out{rank}.npy"))
        assert got.shape == (n, 97) and (got == ref).all()
