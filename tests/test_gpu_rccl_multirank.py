"""BASELINE configs[3] readiness: the data-parallel step over RCCL (torch.distributed backend "nccl" on ROCm), ONE GPU PER
RANK.  Skipped below two visible GPUs -- the build's 1-GPU leases run the same worker over gloo with both ranks on cuda:0
(tests/test_gpu_distributed.py); on the first multi-GPU node this file validates the collective sequence instead of
discovering it (scripts/run.py:81-93, models/robust_e_nerf.py:63-66,916-919):

* shard-gradient sum over RCCL == the single-process gradient of the whole batch (with gradient accumulation);
* a rank without a single sample issues the same collective sequence as its peer;
* replicas (parameters, Adam moments, occupancy grids) bit-identical after K steps with refreshes / batch-size changes;
* a sharded evaluation render (row bands all-gathered over RCCL) == the single-rank render.
"""
import os

import pytest
import torch
import torch.multiprocessing as mp

import test_gpu_distributed as two_rank

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL multi-rank test needs >= 2 GPUs (one per rank)")]


def test_two_ranks_over_rccl_one_gpu_each(tmp_path):
    port = two_rank._free_port()
    mp.spawn(two_rank._worker, args=(port, str(tmp_path), "nccl", True), nprocs=two_rank.WORLD, join=True)
    got = torch.load(os.path.join(tmp_path, "r0.pt"))
    assert got["backend"] == "nccl" and got["world"] == two_rank.WORLD and got["device"] == "cuda:0"
    two_rank.check_two_rank_results(got)
