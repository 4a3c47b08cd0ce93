"""World-size-2 CPU rehearsal (gloo) of the multi-GPU path: tile-aligned range shards, one
gather of codebook records to rank 0, result identical to the unsharded encode.  The per-rank
compute is the oracle here (CPU box, test only); on GPUs it is Encoder.encode(begin, count)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outfile, force_all_gather):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fic_amd
    from fic_amd import sharding
    from oracle import fic_oracle as fo
    sharding._USE_ALL_GATHER = bool(force_all_gather)     # exercise the backend-without-gather fallback too
    g = np.load(os.path.join(ROOT, "tests", "golden", "lena64.npy"))
    B, wK, tile = 4, 29, 64
    argb = fo.gray_to_argb(g)
    spans = fic_amd.shard_spans(256, tile, world)
    b, c = spans[rank]
    e = fo.encode_gray(argb, 64, 64, B, wK, n_iso=8, r0=b, r1=b + c)
    q = fo.quantise_gray(e["info"])
    res = {"idx_local": e["info"][None, :, 0].astype(np.int32), "a": e["info"][None, :, 1].copy(),
           "b": e["info"][None, :, 2].copy(), "iso": e["iso"][None], "qrows": q[None]}
    rec = torch.from_numpy(fic_amd.pack_records(res, b, c))
    full = fic_amd.gather_records(rec, spans, None, 0)
    if rank == 0:
        np.save(outfile, full.numpy())
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,force_all_gather", [(2, False), (3, False), (2, True)])
def test_sharded_gather_equals_unsharded(tmp_path, world, force_all_gather, oracle, lena64):
    out = str(tmp_path / "full.npy")
    mp.spawn(_worker, args=(world, _free_port(), out, force_all_gather), nprocs=world, join=True)
    import fic_amd
    got = fic_amd.unpack_records(np.load(out))
    e = oracle.encode_gray(oracle.gray_to_argb(lena64), 64, 64, 4, 29, n_iso=8)
    assert (got["idx_local"][0] == e["info"][:, 0].astype(np.int32)).all()
    assert (got["a"][0].view(np.uint32) == e["info"][:, 1].view(np.uint32)).all()
    assert (got["b"][0].view(np.uint32) == e["info"][:, 2].view(np.uint32)).all()
    assert (got["iso"][0] == e["iso"]).all()
    assert (got["qrows"][0] == oracle.quantise_gray(e["info"])).all()
