#!/usr/bin/env python
"""Benchmark of the hot path: one training step = event batch -> poses -> rays -> sampling ->
hash grid -> fused MLPs -> compositing -> event loss -> backward -> (all-reduce) -> Adam.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver through torch.distributed.run (one process per GPU, RCCL).
Workload = BASELINE.json configs[1] as SURVEY.md 8 defines config B: synthetic 'ficus'-like event stream, R = 65 536
rays PER RENDER x 128 samples, i.e. 65 536 events per step whose two renders (start / end timestamps) run as one
131 072-ray pass (n = 16 777 216 samples per kernel launch), fp32, default synthetic.yaml NGP model.
`--events 32768` is the other reading (65 536 rays per STEP, n = 8 388 608 per launch; profiles/rNN_bench_half.json).
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant
kernel (HIP-event timed inside the timed region) and `cpu_baseline` (the CPU oracle on a bounded
sample of the same workload, rank 0, N = 1 only).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0              # dense bf16 (MI355X_MICROARCH.md; not the 2:1-sparsity figure)
MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (= f32 vector peak)
# algorithmic bytes per differentiable sample (DESIGN.md "Roofline"):
BYTES = {"hashgrid_fwd": 1024 + 12, "hashgrid_bwd": 2048 + 12, "hashgrid_bwd_binned": 2048 + 12}
# algorithmic flops per sample of the fused MLPs, C = 1 (DESIGN.md "Roofline"): 2 x MACs of
# base 32-64-16 + head 31-64-64-1 forward; backward = data + weight gradients = 2 x forward (no recompute counted)
MLP_MACS = 32 * 64 + 64 * 16 + 31 * 64 + 64 * 64 + 64
VANILLA_MACS = 593152            # 63-256x4-(+63)-256x3, sigma, bottleneck, 283-128-1
FLOPS = {"mlp_fwd": 2 * MLP_MACS, "mlp_bwd": 4 * MLP_MACS, "mlp_fwd_save": 2 * MLP_MACS, "mlp_bwd_saved": 4 * MLP_MACS,
         "mlp_fwd_x": 2 * MLP_MACS, "mlp_bwd_x": 4 * MLP_MACS, "dense_fwd": 2 * VANILLA_MACS,
         "dense_bwd_data": 2 * VANILLA_MACS, "dense_bwd_weight": 2 * VANILLA_MACS,
         # arch mlp, fused field (csrc/ren_vfield.hip): all twelve layers; backward-data has no first layer and only the
         # bottleneck columns of the colour hidden layer
         "vfield_fwd": 2 * VANILLA_MACS, "vfield_bwd": 2 * (VANILLA_MACS - 63 * 256 - 63 * 256 - 27 * 128),
         "vfield_bwd_weight": 2 * VANILLA_MACS}
# compulsory HBM bytes per sample of the hash-grid parameter gradient: 128 B of feature gradients + 12 B of
# sample stream; the 50 MB table pass per launch comes on top.  SURVEY 8(d)'s 2 048 B is the atomic-RMW traffic of the
# scatter-add formulation, kept as `achieved`'s numerator because it is the survey's figure; both are printed.
COMPULSORY_BYTES = {"hashgrid_bwd_binned": 128 + 12, "hashgrid_bwd": 128 + 12, "hashgrid_fwd": 128 + 12}
ROUND = "r06"
PMC_TRAFFIC = os.path.join(REPO, "profiles", f"{ROUND}_pmc_traffic.json")


def kernels_digest():
    """content hash of csrc/ + flags (robust_e_nerf_amd/build.py): PMC passes are valid for the kernels they were taken on"""
    from robust_e_nerf_amd import build
    return build.source_stamps(with_compiler=False)[1]


def pmc_traffic(call, args):
    """HBM bytes per launch of `call` from the committed rocprofv3 PMC passes (tools/pmc_traffic.py), or None when they
    were taken on a different workload OR on different kernels (the file carries the digest of the sources it measured:
    a stale file yields `"traffic": null`, never the numbers of old kernels)."""
    path = PMC_TRAFFIC
    if args.arch == "mlp":
        path = PMC_TRAFFIC.replace(".json", "_arch_mlp_bf16.json" if args.mlp_bf16 else "_arch_mlp.json")
    try:
        t = json.load(open(path))
    except OSError:
        return None
    if t.get("kernels_digest") != kernels_digest():
        return None
    w = t.get("workload", {})
    same = (w.get("events") == args.events and w.get("samples") == args.samples and
            w.get("sampler") == args.sampler and float(w.get("loss_grad", 0.0)) == float(args.loss_grad) and
            w.get("arch", "ngp") == args.arch and bool(w.get("mlp_bf16", False)) == (bool(args.mlp_bf16) if args.arch == "mlp" else False))
    c = t.get("calls", {}).get(call)
    return c["hbm_bytes_per_launch"] if (same and c) else None


def synthetic_scene(n_poses=2001, seed=0, hard=False):
    """radius-4 orbit with slow z oscillation looking at the origin, 1 ms pose spacing,
    346x260 camera, K = [[480,0,172.5],[0,480,129.5],[0,0,1]].  hard (BASELINE configs[2], "non-uniform motion"): the
    same path traversed at non-uniform speed -- the path parameter is a smooth monotone warp of time (speed varies by
    a factor of ~4 along the orbit) while the pose samples stay 1 ms apart."""
    k = np.arange(n_poses)
    u = k / (n_poses - 1)
    if hard:
        u = u + 0.6 / (2 * np.pi * 3) * np.sin(2 * np.pi * 3 * u)      # du/dt = 1 + 0.6 cos(6 pi t) > 0
    ang = 2 * np.pi * u * 1.5
    pos = np.stack([4 * np.cos(ang), 4 * np.sin(ang), 0.6 * np.sin(3 * ang)], -1)
    fwd = -pos / np.linalg.norm(pos, axis=-1, keepdims=True)
    right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
    right /= np.linalg.norm(right, axis=-1, keepdims=True)
    down = np.cross(fwd, right)
    Rm = np.stack([right, down, fwd], -1)
    from scipy.spatial.transform import Rotation
    quat = Rotation.from_matrix(Rm).as_quat().astype(np.float32)
    ts = (k * 1_000_000).astype(np.int64)
    K = np.array([[480.0, 0, 172.5], [0, 480.0, 129.5], [0, 0, 1]], np.float32)
    return ts, pos.astype(np.float32), quat, np.linalg.inv(K).astype(np.float32)


E_AABB = (0.5, -2.1, 0.6, 2.0, -0.6, 1.6)          # configs/train/mocap-desk2.yaml:38-39


def synthetic_scene_e(n_poses=2001):
    """BASELINE configs[4] / SURVEY 8(d) config E: a hand-held orbit inside the mocap room (AABB above), 640x480, f = 500."""
    ts, pos, quat, _ = synthetic_scene(n_poses)
    centre = np.array([1.25, -1.35, 1.1], np.float32)
    pos = (pos * np.array([0.22, 0.22, 0.3], np.float32) + centre).astype(np.float32)
    K = np.array([[500.0, 0, 319.5], [0, 500.0, 239.5], [0, 0, 1]], np.float32)
    return ts, pos, quat, np.linalg.inv(K).astype(np.float32)


def synthetic_events(B, t_end_ns, seed, width=346, height=260):
    g = np.random.default_rng(seed)
    px = np.stack([g.integers(0, width, B), g.integers(0, height, B)], -1).astype(np.float32)
    end = g.integers(20_000_000, t_end_ns, B).astype(np.int64)
    delta = np.exp(g.uniform(np.log(2e5), np.log(2e7), B)).astype(np.int64)
    pol = g.random(B) < 0.5
    return dict(position=px, start_ts=end - delta, end_ts=end, num_pos=pol.astype(np.int64),
                num_neg=(~pol).astype(np.int64), u_ts_diff=np.ones(B), u_diff_start=g.uniform(0, 1, B),
                u_grad=g.uniform(0, 1, B))


def ball_binary(res, radius, aabb):
    lo, hi = np.array(aabb[:3]), np.array(aabb[3:])
    g = np.stack(np.meshgrid(*[np.arange(res)] * 3, indexing="ij"), -1)
    c = (g + 0.5) / res * (hi - lo) + lo
    return (np.linalg.norm(c, axis=-1) < radius).astype(np.uint8).reshape(-1)


def host_cpu():
    """(logical CPUs of the host, CPU model string) -- SURVEY 8(d): stated with the CPU baseline"""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return os.cpu_count(), model


def cpu_baseline(args, scene, params_cpu, table_seed):
    """The CPU oracle (kind 'port': the reference itself cannot execute on a CPU) on a bounded sample: config A =
    2 x 2048 rays x 64 samples, forward + backward + Adam; 3 warm-up + 10 timed steps, median and best."""
    from oracle import field as ofield, hashgrid, step as ostep
    n_cpu, model = host_cpu()
    torch.set_num_threads(min(n_cpu, 32))                # the oracle's ops stop scaling (and start thrashing) past ~32 threads
    tab_ts, tab_pos, tab_quat, Kinv = scene
    spec = hashgrid.make_spec()
    p = {k: v.clone().requires_grad_() for k, v in params_cpu.items()}
    bk = torch.tensor([0.5413]).requires_grad_()
    B, S = 2048, 64
    WARM, TIMED = 3, 10
    opt = torch.optim.Adam([{"params": list(p.values()), "weight_decay": 1e-6}, {"params": [bk]}], lr=0.01)
    cfg = ostep.SceneCfg(sampler="uniform", n_uniform=S)
    T = torch.from_numpy
    times, n_samples = [], 0
    for it in range(WARM + TIMED):
        ev = synthetic_events(B, int(tab_ts[-1]), seed=100 + it)
        batch = ostep.EventBatch(*(T(ev[k]) for k in ("position", "start_ts", "end_ts", "num_pos", "num_neg",
                                                       "u_ts_diff", "u_diff_start")), T(np.zeros(B)))
        g = torch.Generator().manual_seed(it)
        t0 = time.perf_counter()
        loss, aux = ostep.training_forward(
            batch, p, spec, cfg, Kinv=T(Kinv), tab_ts=T(tab_ts), tab_pos=T(tab_pos), tab_quat=T(tab_quat),
            p2n_raw=torch.tensor(0.5413), neg_ct=torch.tensor(0.25), tau_raw=torch.tensor(0.0, dtype=torch.float64),
            tau_max=torch.tensor(1e5), bkgd_raw=bk, binary=None, jitter_start=torch.rand(B, generator=g),
            jitter_end=torch.rand(B, generator=g))
        opt.zero_grad()
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        n_samples = aux["n_start"] + aux["n_end"]
        if sum(times) > 60.0 and it >= WARM + 2:           # a slow host: keep the default bench run within minutes
            break
    steady = times[WARM:]
    best, med = min(steady), float(np.median(steady))
    return {"value": 2 * B / med, "value_best": 2 * B / best, "unit": "rays/s", "samples_per_sec": n_samples / med,
            "cores": torch.get_num_threads(), "host_cpu_count": n_cpu, "host_cpu_model": model, "kind": "port",
            "sample": f"{WARM} warm-up + {len(steady)} timed steps of 2x{B} rays x {S} samples (config A), fwd+bwd+Adam, "
                      f"{sum(times):.1f} s of CPU work, median step {med:.2f} s, best {best:.2f} s, "
                      f"{torch.get_num_threads()} torch threads on {n_cpu} logical CPUs ({model})"}


def main():
    # stdout carries exactly ONE line, the JSON result: native libraries that write to file descriptor 1 (RCCL prints a version
    # banner there when its process group goes away) are sent to stderr for the life of the process
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--events", type=int, default=65536,
                    help="events per step per GPU (2 rays each): 65536 = R 65536 rays per render of BASELINE configs[1] "
                         "(SURVEY 8 config B); 32768 = 65536 rays per step")
    ap.add_argument("--samples", type=int, default=128, help="samples per ray (uniform sampler)")
    ap.add_argument("--sampler", default="uniform", choices=["uniform", "occgrid"])
    ap.add_argument("--workload", default="b", choices=["b", "e"],
                    help="b: BASELINE configs[1] (default).  e: the settings of configs[4] / configs/train/mocap-desk2.yaml on "
                         "synthetic events: sphere contraction (rays march near -> far), 256^3 occupancy grid, cone angle 0.004, "
                         "near 0.05 / far 3.0, l_diff + l_grad, C_p and tau trainable, no background parameter, 640x480 camera")
    ap.add_argument("--arch", default="ngp", choices=["ngp", "mlp"],
                    help="ngp = hash grid (BASELINE configs[1], default); mlp = frequency encoding + 8x256 MLP "
                         "(10.4 KB of saved activations per sample: use --events 8192 or less)")
    ap.add_argument("--loss-grad", type=float, default=0.0,
                    help="weight of the log-intensity-gradient loss (adds a third render with d/dt; 1e-3 in the real-data configs)")
    ap.add_argument("--mlp-bf16", action="store_true",
                    help="BASELINE configs[2] numerics: bf16-rounded linear inputs/weights, fp32 accumulate + composite")
    ap.add_argument("--mlp-precision", default="highest", choices=["highest", "high", "medium"],
                    help="the YAMLs' float32_matmul_precision: highest = every MLP product to fp32 round-off (default, the "
                         "BASELINE configs[1] line); high = each fp32 value as two bf16 pieces, three products; "
                         "medium = --mlp-bf16")
    ap.add_argument("--prefetch", action="store_true",
                    help="run the next step's batch / ray / sample-count front on a side stream (Trainer.prefetch): the host no "
                         "longer enqueues it on the critical path; ordered after the current step's backward since round 4 "
                         "(profiles/NOTES.md), measured worth nothing either way: 12.03 vs 12.04 ms/step in round 2, 3.21 vs 3.11 "
                         "with the occupancy sampler now")
    ap.add_argument("--fwd-chunks", type=int, default=8,
                    help="> 1: hash encoding and MLP forward of alternate sample chunks on two HIP streams")
    ap.add_argument("--bwd-chunks", type=int, default=1,
                    help="> 1: MLP backward of sample chunk k + 1 beside the binned hash-grid scatter of chunk k on two HIP streams, one "
                         "accumulate at the end (RenderCfg.bwd_chunks)")
    ap.add_argument("--bwd-mlp-cus", type=int, default=192, help="CUs of the MLP backward kernels while a scatter runs beside them")
    ap.add_argument("--mlp-kernels", default="x", choices=["x", "f32"],
                    help="x = split-bf16 matrix-core MLP kernels at fp32 accuracy (default), f32 = exact f32-MFMA kernels")
    ap.add_argument("--hard", action="store_true",
                    help="BASELINE configs[2] 'lego hard' shape: non-uniform camera speed, C_p and tau trainable and mis-initialised "
                         "(ratio 1.0, tau 0.999 tau_max), l_grad on (1e-3 unless --loss-grad is given); combine with --mlp-bf16")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --events per GPU whatever N (default).  strong: the reference's semantics (a fixed GLOBAL budget, "
                         "train_eff_ray_sample_batch_size // num_gpus, robust_e_nerf.py:63-66): --events is the global batch, "
                         "every rank takes events // N")
    ap.add_argument("--dp-compress", default=None, choices=["bf16"],
                    help="N > 1: parameter gradients cross the links as bfloat16 (the aux scalars stay fp32)")
    ap.add_argument("--no-dp-overlap", action="store_true",
                    help="N > 1: one all-reduce of the packed gradient buffer after the backward pass instead of reducing the "
                         "fine levels' slice beside the coarse levels' scatter")
    ap.add_argument("--save-activations", type=int, default=-1, choices=[-1, 0, 1],
                    help="1: the training forward stores the hidden MLP activations (768 B/sample), 0: the backward recomputes "
                         "them, -1 (default): recompute with the x kernels, save with the exact-f32 kernels")
    ap.add_argument("--grad-sampling", default="auto", choices=["auto", "begun", "early", "inorder"],
                    help="l_grad term, where the third render's samples are placed: auto = what Trainer.step does "
                         "(Trainer.grad_sampling_mode); begun = front (rays, march count) on the side "
                         "stream before the l_diff pass, the rest beside its backward (Trainer.step); early = all of it beside the "
                         "l_diff backward (experiment only: wrong rays observed, profiles/NOTES.md); inorder = after the l_diff "
                         "backward on the main stream (as before round 4)")
    ap.add_argument("--device-counts", default="auto", choices=["auto", "off"],
                    help="occupancy sampler: sample counts stay on the device (RenderCfg.device_counts); off = the reference's host reads")
    ap.add_argument("--graph", default="auto", choices=["auto", "off"],
                    help="auto: occupancy-sampled steps with device-side counts go through Trainer.step, which captures a repeating "
                         "step in a hipGraph and replays it as one launch (Trainer._graph_step); off: every launch from Python")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    import faulthandler
    faulthandler.enable()
    faulthandler.dump_traceback_later(900, exit=False, file=sys.stderr)

    def log(*a):
        if args.verbose:
            print(f"[bench {time.perf_counter():.1f}]", *a, file=sys.stderr, flush=True)

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("for --gpus N > 1 launch with torch.distributed.run --nproc-per-node N")
    import torch.distributed as dist
    pg = None
    # REN_BENCH_DIST=gloo:shared-gpu runs the N-rank code path on ONE device over gloo (a self-test of the launch
    # contract on a 1-GPU box; numbers mean nothing).  Normal runs: one rank per GPU over RCCL.
    selftest = os.environ.get("REN_BENCH_DIST", "") == "gloo:shared-gpu"
    if selftest:
        local_rank = 0
    # REN_BENCH_DIST=nccl:single-rank: ONE rank with an RCCL process group whose trainer takes the data-parallel code path
    # (early fine-level slice, packed aux block, asynchronous all-reduces on RCCL's stream); a smoke test of the RCCL
    # calls on a 1-GPU box -- the reductions are identities, the Adam scale is that of two ranks.
    rccl_single = os.environ.get("REN_BENCH_DIST", "") == "nccl:single-rank"
    dp_world = 2 if rccl_single else world
    if world > 1 or rccl_single:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29519")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if rccl_single:
            dist.init_process_group(backend="nccl", rank=0, world_size=1)
        else:
            dist.init_process_group(backend="gloo" if selftest else "nccl")
    dev = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)

    from robust_e_nerf_amd import engine, ops          # the oracle is imported by the cpu_baseline leg only

    if args.workload == "e":
        args.sampler = "occgrid"
        if args.loss_grad == 0.0:
            args.loss_grad = 1e-3
    if args.hard and args.loss_grad == 0.0:
        args.loss_grad = 1e-3
    scene = synthetic_scene_e() if args.workload == "e" else synthetic_scene(hard=args.hard)
    tab_ts, tab_pos, tab_quat, Kinv = scene
    T = torch.from_numpy
    # field parameters: torch nn.Linear default init, hash table U(+-0.1) ("trained-like": random
    # operands keep clocks honest, MI355X_MICROARCH.md DVFS note)
    gen = torch.Generator().manual_seed(0)

    def lin(o, i):
        b = 1 / math.sqrt(i)
        return (torch.rand(o, i, generator=gen) * 2 - 1) * b, (torch.rand(o, generator=gen) * 2 - 1) * b
    p = {}
    p["base.w0"], p["base.b0"] = lin(64, 32)
    p["base.wo"], p["base.bo"] = lin(16, 64)
    p["head.w0"], p["head.b0"] = lin(64, 31)
    p["head.w1"], p["head.b1"] = lin(64, 64)
    p["head.wo"], p["head.bo"] = lin(1, 64)
    p["hash"] = (torch.rand(ops.make_grid_desc()[1], generator=gen) * 2 - 1) * 0.1

    aabb = (-1.5, -1.5, -1.5, 1.5, 1.5, 1.5)
    if args.mlp_precision == "medium":
        args.mlp_bf16 = True
    precision = "medium" if args.mlp_bf16 else args.mlp_precision
    high = precision == "high" and args.mlp_kernels == "x"      # (the exact-f32 kernels run "high" at fp32 accuracy)
    cfg = engine.RenderCfg(aabb=aabb, sampler=args.sampler, n_uniform=args.samples, mlp_bf16=args.mlp_bf16,
                           mlp_precision=precision, mlp_kernels=args.mlp_kernels, fwd_chunks=args.fwd_chunks,
                           save_activations=None if args.save_activations < 0 else bool(args.save_activations))
    cfg.dp_overlap = not args.no_dp_overlap
    cfg.dp_compress = args.dp_compress
    cfg.bwd_chunks, cfg.bwd_mlp_cus = args.bwd_chunks, args.bwd_mlp_cus
    if args.workload == "e":
        aabb = E_AABB
        cfg = engine.RenderCfg(aabb=aabb, contraction_type=ops.UN_BOUNDED_SPHERE, occ_res=(256,) * 3, near_plane=0.05,
                               far_plane=3.0, render_step_size=math.sqrt(3) * 1.5 / 1024, cone_angle=0.004,
                               sampler="occgrid", mlp_bf16=args.mlp_bf16, mlp_precision=precision, mlp_kernels=args.mlp_kernels,
                               fwd_chunks=args.fwd_chunks,
                               save_activations=None if args.save_activations < 0 else bool(args.save_activations),
                               dp_overlap=not args.no_dp_overlap, dp_compress=args.dp_compress)
    if args.arch == "mlp":
        from robust_e_nerf_amd import vanilla
        fld = vanilla.VanillaField(dev, 1)
        fld.load({k: v for name, o, i in vanilla.layer_shapes(1)
                  for k, v in zip((name + ".weight", name + ".bias"), lin(o, i))})
        r = vanilla.VanillaRenderer(fld, cfg)
    else:
        fld = engine.NGPField(dev)
        fld.load(p)
        r = engine.Renderer(fld, cfg)
    if args.workload == "e":
        # occupied: a ball of 40 % of the contraction sphere's radius around the room centre (objects on the desk)
        g3 = np.stack(np.meshgrid(*[np.arange(256)] * 3, indexing="ij"), -1)
        r.binary.copy_(T((np.linalg.norm((g3 + 0.5) / 256 - 0.5, axis=-1) < 0.1).astype(np.uint8).reshape(-1)).to(dev))
    elif args.sampler == "occgrid":
        r.binary.copy_(T(ball_binary(128, 0.42, aabb)).to(dev))
    tcfg = engine.TrainCfg(w_grad=args.loss_grad)
    if args.workload == "e":
        tcfg = engine.TrainCfg(w_grad=args.loss_grad, bkgd_is_param=False, train_contrast_threshold=True,
                               train_refractory_period=True)
    p2n0, tau0 = torch.tensor(0.5413), torch.tensor(0.0, dtype=torch.float64)
    if args.hard:                                    # ratio initialised at 1.0, tau at 0.999 tau_max (README.md:103)
        tcfg = engine.TrainCfg(w_grad=args.loss_grad, train_contrast_threshold=True, train_refractory_period=True)
        p2n0 = torch.tensor(math.log(math.expm1(1.0)))
        tau0 = torch.tensor(1e5 * float(torch.logit(torch.tensor(0.999, dtype=torch.float64))), dtype=torch.float64)
    tr = engine.Trainer(r, tcfg, Kinv=T(Kinv), tab_ts=T(tab_ts), tab_pos=T(tab_pos), tab_quat=T(tab_quat),
                        p2n_raw=p2n0, neg_ct=torch.tensor(0.25), tau_raw=tau0, tau_max=torch.tensor(1e5),
                        bkgd_raw=torch.tensor([0.5413]), world_size=dp_world, process_group=pg)
    if args.device_counts == "off":
        tr.device_counts = False
    # the whole step through Trainer.step (what scripts/train.py calls): it replays a captured step when it can
    whole_step = (args.graph == "auto" and args.sampler == "occgrid" and not args.prefetch and
                  args.grad_sampling in ("auto", "inorder") and tr.device_counts_ok())
    if whole_step and args.grad_sampling == "inorder":
        tr.early_grad_sampling = False
    if not whole_step:
        tr.use_graph = False

    B = args.events if args.scaling == "weak" else max(1, args.events // world)
    n_batches = 4                                    # pre-staged in HBM; per-rank seeds (datamodule.py:85-89)
    batches = []
    for b in range(n_batches):
        ev = synthetic_events(B, int(tab_ts[-1]), seed=1 + 1000 * rank + b,
                              **(dict(width=640, height=480) if args.workload == "e" else {}))
        batches.append({k: T(v).to(dev).contiguous() for k, v in ev.items()})
    if whole_step:
        batches = [engine.pack_batch(b) for b in batches]    # every field a view into one buffer: one copy launch per replayed step
    jgen = torch.Generator(device=dev).manual_seed(3 + rank)

    # input pipelining (Trainer.prefetch): the next step's batch, jitters, rays and sample count are produced on a side stream
    # while this step's backward runs -- possible when nothing in that front depends on this step's update (frozen C_p / tau,
    # fixed-S sampler)
    # (occupancy sampler: the early part is the march itself, 0.46 ms of latency chain per 131 k rays; measured 3.17 vs 3.13
    # ms/step with and without -- the march finds no free registers beside the backward kernels, so it stays opt-in)
    can_prefetch = args.prefetch and not tcfg.train_contrast_threshold and not tcfg.train_refractory_period
    staged = {}

    def draw(i):
        with torch.cuda.stream(tr.side_stream if can_prefetch else torch.cuda.current_stream()):
            return batches[i % n_batches], ops.uniform(2 * B, 1234 + rank, i, device=dev), None   # both renders' jitters (library Philox stream)

    def one_step(i):
        if whole_step:
            # jitters drawn straight into the captured step's input buffers once there is one (no copy launches)
            b = batches[i % n_batches]
            gi = tr.graph_inputs(b, jbuf[0], jbuf[1])
            j0, j2 = (gi[1], gi[2]) if gi is not None else (jbuf[0], jbuf[1])
            ops.uniform(2 * B, 1234 + rank, i, device=dev, out=j0)
            if j2 is not None:
                ops.uniform(B, 4321 + rank, i, device=dev, out=j2)
            loss, aux = tr.step(b, j0, None, jitter_grad=j2)
            if args.loss_grad > 0:
                aux = dict(aux, n=aux["n"] + aux["grad"]["n"], rays=aux["rays"] + aux["grad"]["rays"], n_main=aux["n"])
            return loss, aux
        b, j0, j1 = staged.pop(i) if i in staged else draw(i)
        # the third render's jitter is drawn BEFORE the l_diff pass is enqueued: its samples are then placed beside that pass's
        # backward (Trainer.grad_loss_forward_backward(early=True), what Trainer.step does)
        j2 = ops.uniform(B, 4321 + rank, i, device=dev) if args.loss_grad > 0 else None
        gmode = tr.grad_sampling_mode() if args.grad_sampling == "auto" else args.grad_sampling
        if args.loss_grad > 0 and gmode == "begun":
            tr.begin_grad_sampling(b, j2)
        loss, aux = tr.forward_backward(b, j0, j1)
        if can_prefetch:
            staged[i + 1] = draw(i + 1)
            tr.prefetch(*staged[i + 1])
        if args.loss_grad > 0:
            lg, aux_g = tr.grad_loss_forward_backward(b, j2, early={"inorder": False, "begun": True, "early": "all"}[gmode])
            loss = loss + lg
            aux = dict(aux, n=aux["n"] + aux_g["n"], rays=aux["rays"] + aux_g["rays"], n_main=aux["n"])
        tr.optimizer_step()
        return loss, aux

    jbuf = (torch.empty(2 * B, device=dev), torch.empty(B, device=dev) if args.loss_grad > 0 else None)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if tr.sync is not None:
        # RCCL sets up its channels / buffers lazily on the first collectives of each size class: do that here, not in the
        # driver's (possibly short) warm-up, with the step's own call pattern on the real gradient buffer (all zeros)
        f_ = r.field
        for _ in range(2):
            tr.sync.early(f_.grad_all, f_.grad_all.numel() // 3, 2 * (f_.grad_all.numel() // 3))
            tr.sync.finish(f_.grad_all)
        tr.sync.reset_count()
        torch.cuda.synchronize()
    if whole_step:
        # set-up, not warm-up: the first steps read their sample counts on the host, learn the capacities and capture the
        # step; the W warm-up steps and the K timed steps below then are what a training run spends its time in -- replays
        for i in range(16):
            before = (tr.graph_replays, tr.graph_captures)
            one_step(-1 - i)
            if tr.graph_replays == before[0] + 1 and tr.graph_captures == before[1] and i >= 3:
                break
    log("setup done")
    for i in range(args.warmup):
        one_step(i)
        if args.verbose:
            torch.cuda.synchronize()
            log("warmup step", i)
    barrier()
    tr._replays_before = getattr(tr, "graph_replays", 0)
    ops.profile_start()
    t0 = time.perf_counter()
    n_samples = n_main = 0
    for i in range(args.steps):
        loss, aux = one_step(i)
        n_samples += aux["n"]
        n_main += aux.get("n_main", aux["n"])        # samples seen by the profiled (non-tangent) kernel families
    barrier()
    dt = time.perf_counter() - t0
    prof = ops.profile_stop()
    log("timed region done", dt)
    graph_replays, n_main_timed = getattr(tr, "graph_replays", 0), n_main
    roof_note = None
    if graph_replays - tr._replays_before > 0:
        # the timed steps were hipGraph replays: no launch went through the Python wrappers, so there were no per-launch HIP
        # events.  The dominant call is timed over a few EAGER steps of the same workload right behind the timed region.
        tr.use_graph = False
        ops.profile_start()
        n_main, k_prof = 0, min(5, args.steps)
        for i in range(k_prof):
            _, aux_p = one_step(args.steps + i)
            n_main += aux_p.get("n_main", aux_p["n"])
        prof = ops.profile_stop()
        roof_note = f"launch times: HIP events over {k_prof} eager steps behind the timed region (the {args.steps} timed steps are hipGraph replays)"
        prof_steps = k_prof
    else:
        prof_steps = args.steps
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
        ns = torch.tensor([n_samples], device=dev, dtype=torch.float64)
        dist.all_reduce(ns)
        n_samples = float(ns)
    rays = (3 if args.loss_grad > 0 else 2) * B * args.steps * world

    # N > 1, default (weak) run: the reference's own semantics as a second measurement in the same JSON line -- a fixed
    # GLOBAL batch (train_eff_ray_sample_batch_size // num_gpus, robust_e_nerf.py:63-66): every rank takes events // N.
    # One driver run per N then yields both curves.
    strong = None
    if world > 1 and args.scaling == "weak" and not rccl_single:
        B_weak = B
        B = max(1, args.events // world)
        batches = []
        for b in range(n_batches):
            ev = synthetic_events(B, int(tab_ts[-1]), seed=1 + 1000 * rank + b,
                                  **(dict(width=640, height=480) if args.workload == "e" else {}))
            batches.append({k: T(v).to(dev).contiguous() for k, v in ev.items()})
        staged.clear()
        for i in range(args.warmup):
            one_step(i)
        barrier()
        t0s = time.perf_counter()
        ns_s = 0
        for i in range(args.steps):
            _, aux_s = one_step(i)
            ns_s += aux_s["n"]
        barrier()
        dts = torch.tensor([time.perf_counter() - t0s, float(ns_s)], device=dev, dtype=torch.float64)
        tmax = dts[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(dts[1:])
        rays_s = (3 if args.loss_grad > 0 else 2) * B * args.steps * world
        strong = {"scaling": "strong", "events_per_step_global": B * world, "events_per_step_per_gpu": B,
                  "value": rays_s / float(tmax), "unit": "rays/s", "ms_per_step": float(tmax) / args.steps * 1e3,
                  "mlp_samples_per_sec": float(dts[1]) / float(tmax), "steps": args.steps, "warmup": args.warmup}
        B = B_weak

    if "hashgrid_bwd_binned_finish" in prof:
        # the phased call of the chunked backward (begin, one scatter per chunk -- timed on the second stream, i.e. while it
        # shares the chip with the MLP backward --, finish) counts as ONE launch of the C-ABI call per backward pass
        calls = prof["hashgrid_bwd_binned_finish"][0]
        total = sum(prof.pop(k)[1] for k in ("hashgrid_bwd_binned_begin", "hashgrid_bwd_binned_scatter", "hashgrid_bwd_binned_finish"))
        c0, m0 = prof.get("hashgrid_bwd_binned", (0, 0.0))
        prof["hashgrid_bwd_binned"] = (c0 + calls, m0 + total)
    if rank == 0:
        kern = {k: {"launches": c, "avg_ms": ms / max(c, 1)} for k, (c, ms) in prof.items()}
        dom = max(list(BYTES) + list(FLOPS), key=lambda k: prof.get(k, (0, 0.0))[1])
        c, ms = prof.get(dom, (1, 1e-9))                 # degenerate runs (no sample at all) launch none of them
        c, ms = max(c, 1), max(ms, 1e-9)
        # this rank's samples per launch of the dominant kernel (chunked calls: one launch per chunk)
        samples_per_launch = n_main / (prof_steps if dom.startswith("dense") else max(c, 1))
        if dom in BYTES:
            achieved = BYTES[dom] * samples_per_launch / (ms / c * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dom, args),
                    "algorithmic_bytes_per_sample": BYTES[dom], "avg_launch_ms": ms / c,
                    # the survey's figure prices the scatter-add formulation; what this gradient MUST move is far less:
                    "compulsory_bytes_per_sample": COMPULSORY_BYTES[dom],
                    "compulsory_gbs": (COMPULSORY_BYTES[dom] * samples_per_launch + 2 * 4 * fld.n_table) / (ms / c * 1e-3) / 1e9
                    if args.arch == "ngp" else None}
        else:
            # dense_* families: one launch per layer, so price the whole family per step
            per = (ms / prof_steps) if dom.startswith("dense") else (ms / c)
            # matrix-core peak of the path that ran: exact f32 MFMA, or bf16 MFMA with 6 (split-bf16, fp32 accuracy)
            # or 1 (plain bf16) hardware multiply-adds per algorithmic one
            if args.mlp_kernels == "f32" and not args.mlp_bf16:
                peak, terms, pipe = MFMA_F32_PEAK_TFLOPS, 1, "v_mfma_f32_32x32x2_f32"
            else:
                peak, terms, pipe = MFMA_BF16_PEAK_TFLOPS, (1 if args.mlp_bf16 else 3 if high else 6), "v_mfma_f32_32x32x16_bf16"
            algorithmic = FLOPS[dom] * samples_per_launch / (per * 1e-3) / 1e12
            achieved = algorithmic * terms
            roof = {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": peak,
                    "unit": "TFLOP/s", "frac": achieved / peak, "traffic": pmc_traffic(dom, args),
                    "algorithmic_flops_per_sample": FLOPS[dom], "avg_launch_ms": ms / c, "matrix_pipe": pipe,
                    "hardware_multiply_adds_per_algorithmic": terms, "algorithmic_tflops": algorithmic}
            if roof["traffic"]:
                # the fused arch-mlp kernels move their saved copies through HBM: how close the same call is to the HBM peak
                roof["hbm_traffic_gbs"] = roof["traffic"] / (per * 1e-3) / 1e9
                roof["hbm_traffic_frac_of_peak"] = roof["hbm_traffic_gbs"] / HBM_PEAK_GBS
        out = {
            "metric": "train_rays_per_sec", "value": rays / dt, "unit": "rays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            # fp32 storage, products and accumulation everywhere.  The default MLP kernels form every product -- outputs, data
            # gradients AND weight gradients -- as the six-term split on the bf16 matrix cores, i.e. to fp32 round-off
            # (test_matrix_core_mlp_kernels_vs_exact_f32_kernels_at_config_b_size); --mlp-kernels f32 runs them on the f32 MFMA
            "dtype": "bf16-operand/f32-accumulate MLP, f32 elsewhere" if args.mlp_bf16 else
                     "f32 (MLP products: float32_matmul_precision high = two bf16 pieces per value, three products)" if high else "f32",
            "data": "synthetic",
            "mlp_samples_per_sec": n_samples / dt, "mean_samples_per_ray": n_samples / rays,
            "loss": float(loss),
            "config": {"workload": ("BASELINE configs[4] settings (mocap-desk2.yaml: sphere contraction, 256^3 grid, cone 0.004, "
                                    "near/far, C_p + tau trainable) on synthetic events, " if args.workload == "e" else
                                    "BASELINE configs[2] shape (non-uniform motion, C_p + tau trainable, l_grad" +
                                    (", bf16 MLP + fp32 composite" if args.mlp_bf16 else "") + "): synthetic event stream, " if args.hard else
                                    "BASELINE configs[1]: synthetic ficus-like event stream, ") +
                                   f"{B} events/step/GPU = 2 renders x {B} rays x {args.samples} samples, arch {args.arch}, fp32, "
                                   f"{'l_diff + l_grad (3 renders)' if args.loss_grad > 0 else 'l_diff (2 renders)'}, fwd+bwd+Adam; sampler={args.sampler}",
                       "events_per_step_per_gpu": B, "rays_per_step_per_gpu": 2 * B, "sampler": args.sampler,
                       "samples_per_ray": args.samples, "parallelism": f"dp{world}",
                       "collectives_per_step": getattr(tr, "last_collectives", 0), "front_prefetched": bool(can_prefetch),
                       "grad_sampling": (tr.grad_sampling_mode() if args.grad_sampling == "auto" else args.grad_sampling) if args.loss_grad > 0 else None,
                       "mlp_precision": "highest" if precision == "high" and not high else precision,
                       "device_counts": bool(tr.device_counts_ok() and tr.r._spr is not None),
                       "device_count_overflows": getattr(tr, "device_count_overflows", 0),
                       "step_graph": {"replays_in_timed_region": graph_replays - getattr(tr, "_replays_before", 0),
                                      "captures": getattr(tr, "graph_captures", 0),
                                      # every capture is measured against the eager step before it is used (Trainer._capture)
                                      "kept_ms_replay_vs_eager": [[round(v, 3) for v in g_["ms"]] for g_ in tr._graphs.values() if "ms" in g_],
                                      "rejected_captures": sum(tr._graph_bad.values())} if whole_step else None,
                       "fwd_chunks": args.fwd_chunks, "bwd_chunks": args.bwd_chunks,
                       # what torch.distributed actually formed (a mis-launched N-rank run shows here)
                       "dist_world_size": dist.get_world_size() if dist.is_initialized() else 1,
                       "dist_backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else ""))
                       if dist.is_initialized() else None,
                       "gradient_allreduce_bytes_per_step": int(r.field.grad_all.numel() * 4) if dp_world > 1 else 0},
            "roofline": dict(roof, **({"note": roof_note} if roof_note else {})),
            "kernels": kern,
        }
        if strong is not None:
            out["strong_scaling"] = strong
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, scene, {k: v.clone() for k, v in p.items()}, 0)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or rccl_single:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
