#!/usr/bin/env python3
"""bench.py -- headline benchmark of the DiffUS hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no WORLD_SIZE in the environment makes THIS process a launcher: before any GPU call
it starts N child ranks (one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, rendezvous on 127.0.0.1), waits
for them and exits non-zero if any of them failed.  Under torch.distributed.run the ranks already exist and the
process is one of them.  Rank 0 prints the one JSON line.

Metric (BASELINE.json): ray-steps/s, forward + backward, 256^3 volume, 256 rays x
512 steps.  One *step* = one pass of the hot path over one batch of poses:
  forward frame, loss_p = sum(frame_p^2) and the backward of it in ONE call (diffus_render_step_mse: the adjoint-scan
     kernel recomputes the forward per ray anyway, so it also writes the frame and forms dL/dframe = 2 frame on the
     spot; the per-pose loss is summed by the call's per-pose blocks.  --two-pass runs diffus_render_fwd +
     diffus_render_bwd_mse, --unfused-loss diffus_render_fwd + the loss kernel diffus_loss_sumsq + diffus_render_bwd):
     frame, loss, d/d volume, d/d source, d/d directions
  -> touched bricks of the gradient scratch -> the caller's canonical (d0,d1,d2) volume-gradient tensor
     (diffus_gradbuf_flush, mode PERSISTENT: the tensor is kept across steps and bricks the previous step
     wrote but this one does not are cleared, so after every step it IS that step's dense gradient -- the
     64 MiB memset per step this replaces is still available as --memset-grad)
  -> gather of the per-pose losses over ranks (RCCL, N > 1).
Workload at N=1 = BASELINE config 3 (32 poses of the config-2 shape on one GPU;
a single 256x512 frame is only 256 wavefronts, i.e. launch-latency-bound, and is
reported separately as `single_pose`).  Multi-GPU (N > 1) = BASELINE config 4 as SURVEY 8d defines it: STRONG
scaling, --poses-total = 256 poses sharded contiguously over the N ranks (256 / N per GPU), the volume replicated,
the only collective one all_gather of the P losses per step; the line then also carries `weak` (32 poses per GPU),
and `strong.one_gpu` (all 256 poses on rank 0's GPU alone, timed in the same run) with `speedup_vs_one_gpu`.  The
N = 1 line carries that one-GPU leg too (`strong_base` = config 4 at 1 of 8 GPUs).  `--scaling weak|strong` forces
either mode at any N.

After the timed region (never inside it) the step CHECKS ITSELF: frames and per-pose losses of the first and last
pose against the scalar C oracle (`verified`); a mismatch makes the process exit non-zero.

Inputs (volume -- canonical and its converted copy --, poses) are resident in HBM
before the timed region.  The step (diffus_amd.CapturedStep) is issued either eagerly (two C-ABI calls, three kernels)
or as a captured hipGraph replay (the C-ABI never syncs or allocates, so it captures); by default (--issue auto) both are
timed for a few steps after the warm-up and the faster one runs the timed region -- on this stack that is the eager
step: kernels of one stream run back to back, two graph LAUNCHES are ~8.6 us apart (tools/graph_gaps.py).
Kernel durations for the roofline come from HIP events recorded on the launch
stream (torch's current stream, which is the one handed to the C-ABI).

`roofline` leads with what was MEASURED for the dominant kernel: `achieved` / `frac` = HBM-side bytes of the committed
rocprofv3 PMC passes for exactly this workload (profiles/*_pmc_*.json) over its live launch time; the contract's no-reuse
ALGORITHMIC figure (SURVEY 8d bytes per ray-step x ray-steps per launch / launch time) sits beside it under `algorithmic`
-- it is a model, and for cache-resident planar fans it exceeds what crosses the HBM interface.  `bound` names what the
counters say limits the kernel ("hbm" only when the measured traffic is near the HBM rate); `issue_roofline` prices the
kernel's VALU instruction stream against the SIMDs' issue rate, both at the guide's FP32 rate (2 cycles per wave64
instruction) and cost-weighted with the per-instruction costs measured by tools/valu_issue_bench.hip.

`callers` (N = 1 only) times the shapes the reference's notebooks actually run: the REUBEN demo frame
(start > 0, artifacts, splat), a learnable volume (re-converted every step) and poses that move every step.
"""
from __future__ import annotations

import argparse
import ctypes as C
import glob
import json
import os
import signal
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic bytes per ray-step, no-reuse model (SURVEY §8d / DESIGN.md §Roofline), per kernel:
#   fwd      8 corner reads x 4 B + 4 B frame write                         = 36 (nearest: 4 + 4 = 8)
#   bwd scan 4 B gframe read + 8 x 4 B corner re-reads                      = 36 (nearest: 8)
#            (one-pass step: the 4 B are the frame write instead -- it has no dL/dframe to read -- same 36 / 8)
#   scatter  8 corners x (4 B read + 4 B write) atomic RMW on the gradient  = 64 (nearest: 8)
BYTES = {
    "trilinear": {"render_fwd_kernel": 36, "render_bwd_kernel": 36, "scatter_patch_kernel": 64},
    "nearest": {"render_fwd_kernel": 8, "render_bwd_kernel": 8, "scatter_patch_kernel": 8},
}
HBM_PEAK_GBS = 8000.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--poses", type=int, default=32, help="poses per GPU (weak scaling: the per-GPU batch is fixed)")
    ap.add_argument("--poses-total", type=int, default=256,
                    help="strong scaling: poses of the WHOLE job, sharded contiguously over the ranks (BASELINE config 4: 256)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"],
                    help="weak: --poses per GPU whatever N; strong: --poses-total split over the N ranks (SURVEY 8d config 4: "
                         "P=256 fixed, P/k per GPU); auto = config 3 (weak, 32 poses) at N=1 and strong at N>1")
    ap.add_argument("--rays", type=int, default=256)
    ap.add_argument("--samples", type=int, default=512)
    ap.add_argument("--start", type=int, default=0, help="start crop (reference src/renderer.py:237-244)")
    ap.add_argument("--n", type=int, default=256, help="volume edge")
    ap.add_argument("--sampler", default="trilinear", choices=["trilinear", "nearest"])
    ap.add_argument("--no-gvol", action="store_true", help="pose-gradient-only backward")
    ap.add_argument("--layout", default="paired", choices=["paired", "bricked", "canonical"])
    ap.add_argument("--learnable-volume", action="store_true",
                    help="re-convert the volume to the kernels' layout inside every step (a volume an optimiser updates)")
    ap.add_argument("--memset-grad", action="store_true",
                    help="zero the whole canonical gradient tensor every step instead of keeping it persistent "
                         "(diffus_gradbuf_flush mode STORE instead of PERSISTENT)")
    ap.add_argument("--dense-grad", action="store_true",
                    help="hand the gradient back by a dense conversion instead of the touched-brick flush")
    ap.add_argument("--unfused-loss", action="store_true",
                    help="separate loss kernel (diffus_loss_sumsq) + diffus_render_bwd instead of the fused diffus_render_bwd_mse")
    ap.add_argument("--two-pass", action="store_true",
                    help="separate forward launch (diffus_render_fwd + diffus_render_bwd_mse) instead of the one-pass "
                         "diffus_render_step_mse")
    ap.add_argument("--eager", action="store_true", help="issue launches from Python instead of replaying a hipGraph (= --issue eager)")
    ap.add_argument("--issue", default="auto", choices=["auto", "eager", "graph"],
                    help="how a step reaches the GPU: `eager` = the step's launches issued from Python through the C-ABI (two "
                         "calls, three kernels), `graph` = one captured hipGraph replayed per step, `auto` = both are timed for "
                         "a few steps after the warm-up (outside the timed region) and the faster one is used.  On this stack "
                         "consecutive kernels of a stream or of one graph run back to back, but two graph LAUNCHES are ~8.6 us "
                         "apart (tools/graph_gaps.py), so the eager step is the faster one unless the host is slow")
    ap.add_argument("--prewarm-ms", type=float, default=100.0,
                    help="untimed steps for this many milliseconds before the --warmup steps (clocks settle over tens of ms; "
                         "reported as config.prewarm_steps).  0 = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-callers", action="store_true", help="skip the `callers` legs (demo shape, learnable volume, moving poses)")
    ap.add_argument("--alpha", type=float, default=1e-4)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 logic on a single GPU (every rank on cuda:0)")
    ap.add_argument("--sync-gather", action="store_true",
                    help="N > 1: issue the loss all_gather on the compute stream (default: on its own stream, overlapping "
                         "the next step's kernels)")
    ap.add_argument("--gather-every", type=int, default=1,
                    help="N > 1: bucket the per-pose losses of this many steps into one all_gather.  1 (default) = one gather "
                         "per step, what BASELINE config 4 describes; K > 1 is an opt-in optimisation (losses arrive up to K-1 "
                         "steps late) and is reported beside the headline as `bucketed_gather`")
    ap.add_argument("--raw-nccl-gather", action="store_true",
                    help="N > 1: issue the loss all_gather as ncclAllGather through RCCL's C entry point on a communicator of the "
                         "bench's own instead of through torch.distributed.  Measured on a one-rank group (round 4): no faster -- "
                         "RCCL's own enqueue is the ~50 us of host time per collective, c10d adds little (0.0760 against 0.0655 ms "
                         "per step with a gather every step); opt-in, kept for boxes where that differs")
    ap.add_argument("--force-dist", action="store_true",
                    help="debug: initialise torch.distributed even for a single rank, to exercise the N > 1 code path")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work at all: ranks rendezvous (gloo), gather stand-in losses and time empty steps; "
                         "checks the launcher / rendezvous / gather / max-over-ranks logic on a CPU-only box")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the after-the-timed-loop comparison of the step's frames and losses with the CPU oracle")
    ap.add_argument("--no-scaling-legs", action="store_true",
                    help="skip the extra timed legs (one-GPU base of the strong curve, weak figure, bucketed gather)")
    ap.add_argument("--fail-rank", type=int, default=-1, help=argparse.SUPPRESS)   # tests: this rank exits 3 before the rendezvous
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` starts its own N ranks.  Nothing here touches the GPU or imports torch: the device
# count comes from the visibility variables or the KFD topology in sysfs (each child checks its own LOCAL_RANK against what
# HIP shows it and exits non-zero otherwise); the children are ordinary child processes, never an exec of this one.
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpus(environ=None, kfd_nodes="/sys/class/kfd/kfd/topology/nodes"):
    """How many GPUs a child would see, without initialising HIP: the shortest of the *_VISIBLE_DEVICES lists if any is
    set, else the KFD topology nodes that have SIMDs (CPU nodes have simd_count 0).  None = cannot tell."""
    env = os.environ if environ is None else environ
    listed = [len([x for x in env[k].split(",") if x.strip() != ""])
              for k in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES") if env.get(k) is not None]
    n = None
    try:
        n = 0
        for node in os.listdir(kfd_nodes):
            try:
                with open(os.path.join(kfd_nodes, node, "properties")) as fh:
                    props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        n = None
    if listed:
        return min(listed) if n is None else min(min(listed), n)
    return n


def gpu_numa_cpus(local_rank, environ=None, kfd_nodes="/sys/class/kfd/kfd/topology/nodes", drm="/sys/class/drm",
                  numa="/sys/devices/system/node"):
    """The CPUs of the NUMA node the `local_rank`-th GPU hangs off, from sysfs alone (no HIP): KFD GPU nodes in node order ->
    drm_render_minor -> renderD<minor>/device/numa_node -> node<k>/cpulist.  None when any link of that chain is missing, when
    the node is unknown (-1), or when a *_VISIBLE_DEVICES variable reorders the devices (the rank -> node order then is not ours
    to guess).  The per-step loss gather is HOST-bound (~50 us inside RCCL's enqueue): a rank that runs on the far socket pays
    for it on every step."""
    env = os.environ if environ is None else environ
    if any(env.get(k) is not None for k in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")):
        return None
    try:
        gpus = []
        for node in sorted(os.listdir(kfd_nodes), key=lambda x: int(x)):
            with open(os.path.join(kfd_nodes, node, "properties")) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(int(props["drm_render_minor"]))
        with open(os.path.join(drm, "renderD%d" % gpus[local_rank], "device", "numa_node")) as fh:
            k = int(fh.read().strip())
        if k < 0:
            return None
        with open(os.path.join(numa, "node%d" % k, "cpulist")) as fh:
            cpus = set()
            for part in fh.read().strip().split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        return cpus or None
    except (OSError, ValueError, KeyError, IndexError):
        return None


def pin_to_gpu_numa_node(local_rank) -> bool:
    """Restrict this process to the CPUs of its GPU's NUMA node (never widening what it already may use).  Before torch starts
    its threads."""
    cpus = gpu_numa_cpus(local_rank)
    if not cpus or not hasattr(os, "sched_setaffinity"):
        return False
    try:
        allowed = os.sched_getaffinity(0) & cpus
        if len(allowed) < 2:
            return False
        os.sched_setaffinity(0, allowed)
        return True
    except OSError:
        return False


def launch_ranks(args) -> int:
    n = args.gpus
    if args.dist_backend == "nccl" and not args.dry_run:
        have = visible_gpus()
        if have is not None and have < n:
            print(f"bench.py: --gpus {n} but only {have} GPU(s) visible (use --dist-backend gloo to rehearse {n} ranks "
                  f"on one GPU)", file=sys.stderr)
            return 2
    port = _free_port()
    procs = []

    def on_term(signum, frame):                       # SIGTERM to the launcher ends like Ctrl-C: the finally below runs
        raise KeyboardInterrupt

    old_term = signal.signal(signal.SIGTERM, on_term)
    rc = 0
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            # rank 0 owns stdout (the one JSON line); the other ranks' stdout goes to stderr
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                          stdout=None if r == 0 else sys.stderr))
        pending = dict(enumerate(procs))
        while pending:
            for r, p in list(pending.items()):
                code = p.poll()
                if code is None:
                    continue
                del pending[r]
                if code != 0 and rc == 0:          # a rank failed: the others would wait in a collective for ever
                    rc = code if 0 < code < 256 else 1
                    print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                    pending = {}
                    break
            time.sleep(0.05)
    except KeyboardInterrupt:
        rc = rc or 130
    finally:                                          # whatever ended the wait: no rank is left behind
        alive = [p for p in procs if p.poll() is None]
        for p in alive:
            p.terminate()
        t_kill = time.time() + 10
        for p in alive:
            try:
                p.wait(timeout=max(0.0, t_kill - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        signal.signal(signal.SIGTERM, old_term)
    return rc


# ------------------------------------------------------------------------------------------------------------------
def vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def time_events(fn, iters, pre=None):
    """Device time of fn() in ms (mean, median, min): HIP events on the current (launch) stream."""
    import torch
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    for i in range(iters):
        if pre is not None:
            pre()
        e0[i].record()
        fn()
        e1[i].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in zip(e0, e1))
    return {"mean": sum(ts) / len(ts), "median": ts[len(ts) // 2], "min": ts[0]}


def time_wall(fn, iters, warm=3):
    """Wall ms per call, device drained at both ends (what a Python caller waits for)."""
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / iters


def cpu_baseline(budget_rays=64, budget_steps=256):
    """The reference's algorithm (N+1 dense torch.linalg.solve, oracle/dense.py) on
    the host cores, on a bounded sample: config 1 = 64 rays x 256 steps, forward
    only (the dense backward needs ~rays*steps^3 memory)."""
    import torch
    from diffus_amd.phantom import phantom, pose_ring
    from oracle import dense
    from oracle import oracle as orc
    n = 256
    vol = phantom(n)
    src, dirs = pose_ring(n, 32, budget_rays)
    # the GPU box gives one GPU a 16-core share of the host; more threads than that (or than the
    # affinity mask) only oversubscribes the many small LAPACK calls
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    dense.plot_beam_frame_dense(torch.from_numpy(vol), torch.from_numpy(src[0]), torch.from_numpy(dirs[0]),
                                budget_steps, 1e-4, 0)
    dt = time.perf_counter() - t0
    out = {"value": budget_rays * budget_steps / dt, "unit": "ray-steps/s", "cores": torch.get_num_threads(),
           "kind": "port",
           "sample": f"reference algorithm (N+1 dense torch.linalg.solve, oracle/dense.py), forward only, "
                     f"1 pose x {budget_rays} rays x {budget_steps} steps on the 256^3 phantom "
                     f"(BASELINE config 1), {dt:.2f} s wall"}
    # algorithm-matched extra: the O(N) scalar C oracle, one 256x512 pose, trilinear forward
    src2, dirs2 = pose_ring(n, 32, 256)
    orc.build()
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 2.0:
        orc.plot_beam_frame(vol, src2[reps % 32], dirs2[reps % 32], 512, 1e-4, 0, sampler="trilinear")
        reps += 1
    dt2 = time.perf_counter() - t0
    out["cpu_scan_value"] = reps * 256 * 512 / dt2
    out["cpu_scan_note"] = "O(N) running-product C oracle (oracle/diffus_oracle.c), 1 thread, trilinear forward only"
    return out


def workload_key(args, poses=None):
    """What a PMC summary must have been collected on to speak for this run (`poses` = per GPU)."""
    return {"n": args.n, "poses": args.poses if poses is None else poses, "rays": args.rays, "samples": args.samples, "start": args.start,
            "sampler": args.sampler, "layout": args.layout,
            "passes": 2 if (args.two_pass or args.unfused_loss) else 1}


def find_pmc_summary(key):
    """Newest committed profiles/*pmc*.json whose `workload` equals `key` (None if there is none)."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc*.json"))):
        try:
            pm = json.load(open(f))
        except Exception:
            continue
        if pm.get("workload") == key and not pm.get("superseded_by"):   # before/after pairs of one round carry that key
            best = (os.path.relpath(f, ROOT), pm)          # names sort by round: the last match is the newest
    return best


def issue_mix(kernel):
    """Mean issue cycles per VALU instruction of `kernel` (tools/issue_model.py --json, committed under profiles/)."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*issue_model*.json"))):
        try:
            for rec in json.load(open(f)).get("kernels", []):
                if rec.get("short") == kernel:
                    best = dict(rec, source=os.path.relpath(f, ROOT))
        except Exception:
            continue
    return best


def plan_poses(args, world):
    """-> (scaling, P_total, poses per rank).  weak: --poses per GPU whatever N.  strong: --poses-total over the whole
    job, P_total / N per rank.  auto: BASELINE config 3 at N = 1, strong at N > 1."""
    scaling = args.scaling
    if scaling == "auto":
        scaling = "weak" if world == 1 else "strong"
    if scaling == "weak":
        return scaling, args.poses * world, args.poses
    if args.poses_total % world:
        raise SystemExit(f"bench.py: --poses-total {args.poses_total} does not split evenly over {world} ranks")
    return scaling, args.poses_total, args.poses_total // world


def config_label(args, ngpu, P_total=None):
    """Which BASELINE.json config (if any) this run is."""
    if P_total is None:
        P_total = plan_poses(args, ngpu)[1]
    shape2 = (args.rays, args.samples, args.start) == (256, 512, 0)
    if args.n == 256 and shape2:
        if P_total == 256:                         # config 4: 256 poses over k GPUs, k = 1, 2, 4, 8 is its scaling curve
            return "BASELINE config 4" if ngpu == 8 else f"BASELINE config 4 at {ngpu} of 8 GPUs"
        if P_total == 32 and ngpu == 1:
            return "BASELINE config 3"
        if P_total == 32 * ngpu:
            return f"BASELINE config 3 per GPU x {ngpu} GPUs (weak scaling)"
        if P_total == 1 and ngpu == 1:
            return "BASELINE config 2"
    if args.n == 512 and (args.rays, args.samples, args.start) == (512, 1024, 0):
        return f"BASELINE config 5 shape ({P_total // ngpu} poses per 512^3 volume, one volume per GPU)"
    return "custom workload (not a BASELINE.json config)"


# ------------------------------------------------------------------------------------------------------------------
# `callers`: the shapes the reference's notebooks run, N = 1 only
def callers_legs(args, vol, dev):
    import numpy as np
    import torch
    import diffus_amd
    from diffus_amd import CapturedStep
    from diffus_amd import _lib as _lib_mod
    from diffus_amd.phantom import pose_ring
    out = {}
    n = args.n
    # (a) `[DEMO] REUBEN DATA 46` cell 14: 256 rays x 185 samples, start = 40, artifacts=True, alpha = 1e-4, an f64
    #     apex outside the volume, then differentiable_splat onto 256 x 256 (sigma = 1) -- through the drop-in API
    R, S, start = 256, 185, 40
    source = torch.tensor([88.0769, -11.5385, 110.0], dtype=torch.float64) * (n / 256.0)
    dirs = diffus_amd.generate_cone_directions(np.array([0.35, 0.94]), np.radians(52.47), R)
    rend = diffus_amd.UltrasoundRenderer(num_samples=S, attenuation_coeff=1e-4)

    def demo_frame():
        x, y, z, I = rend.plot_beam_frame(volume=vol, source=source, directions=dirs, plot=False, artifacts=True,
                                          start=start, seed=0)
        return diffus_amd.differentiable_splat(x, y, z, I, H=256, W=256, sigma=1)

    def demo_render_only():
        return rend.plot_beam_frame(volume=vol, source=source, directions=dirs, plot=False, artifacts=False, start=start)

    ms_full = time_wall(demo_frame, 20)
    ms_render = time_wall(demo_render_only, 20)
    out["demo_reuben46"] = {
        "call": "UltrasoundRenderer(185, 1e-4).plot_beam_frame(artifacts=True, start=40) + differentiable_splat(256x256, sigma=1), "
                "nearest sampling, f64 apex outside the volume (reference notebooks/[DEMO] REUBEN DATA 46.ipynb cell 14)",
        "ms_per_frame_wall": ms_full, "ms_plot_beam_frame_only_wall": ms_render,
        "ray_steps_per_s": R * S / (ms_full * 1e-3),
        "reference_published": "2.54 s/frame on the authors' laptop CPU at 200 rays x 150 samples (BASELINE.md)"}
    # (a') the drop-in TRAINING step: render_poses -> loss -> .backward() through torch autograd (what a notebook that only
    #      swaps the import runs, e.g. `[NW] alignement` cells 13-14), wall time per step, 32 poses and 1 pose; and the same
    #      with the autograd engine kept on the calling thread (torch.autograd.set_multithreading_enabled(False): the
    #      hand-off to the engine's device thread is ~50 us of a step that has ~100 us of kernels)
    def autograd_step_ms(P, single_thread):
        s_all, d_all = pose_ring(n, 32, args.rays)
        v = vol.detach().clone().requires_grad_(True)
        sp = torch.from_numpy(s_all[:P]).to(dev).requires_grad_(True)
        dp = torch.from_numpy(d_all[:P]).to(dev).requires_grad_(True)

        def step():
            f = diffus_amd.render_poses(v, sp, dp, args.samples, args.alpha, sampler="trilinear")
            (f * f).sum().backward()
            v.grad = None; sp.grad = None; dp.grad = None

        if single_thread:
            with torch.autograd.set_multithreading_enabled(False):
                return time_wall(step, 50, warm=10)
        return time_wall(step, 50, warm=10)

    out["autograd_step"] = {
        "call": "render_poses(volume, sources, directions, 512, 1e-4, sampler='trilinear') -> (f*f).sum().backward(), all three "
                "inputs require grad, 256^3 volume, 256 rays x 512 steps; wall ms per step (host-bound)",
        "poses32_ms": autograd_step_ms(32, False), "poses1_ms": autograd_step_ms(1, False),
        "poses32_single_threaded_engine_ms": autograd_step_ms(32, True),
        "poses1_single_threaded_engine_ms": autograd_step_ms(1, True),
        "round2_ms": {"poses32": 0.307, "poses1": 0.241},
    }
    # the same shape as a batch of 32 captured training steps (forward + loss + backward with the start-crop median)
    src32, dirs32 = pose_ring(n, 32, R)
    s32 = torch.from_numpy(src32).to(dev)
    d32 = torch.from_numpy(dirs32).to(dev)
    for sampler in ("trilinear", "nearest"):
        st = CapturedStep(vol, s32, d32, S, 1e-4, sampler, start=start, layout=args.layout)
        st.capture()
        ms = time_events(st.replay, 30)["median"]
        out["demo_reuben46"][f"batch32_fwd_bwd_{sampler}_ms"] = ms
        out["demo_reuben46"][f"batch32_fwd_bwd_{sampler}_ray_steps_per_s"] = 32 * R * S / (ms * 1e-3)
        del st
    # (b) a learnable volume: the optimiser changes it every step, so the layout conversion is part of the step
    leg = {}
    src_a, dirs_a = pose_ring(n, args.poses, args.rays)
    sa, da = torch.from_numpy(src_a).to(dev), torch.from_numpy(dirs_a).to(dev)
    for layout in ("paired", "bricked", "canonical"):
        st = CapturedStep(vol, sa, da, args.samples, args.alpha, args.sampler, layout=layout, learnable_volume=True)
        st.capture()
        leg[f"{layout}_ms_per_step"] = time_events(st.replay, 30)["median"]
        del st
    # ... or ONE slice of it (the reference's MLP training loop rewrites the slice the fan lies in): only that slice's records
    for layout in ("paired", "bricked"):
        st = CapturedStep(vol, sa, da, args.samples, args.alpha, args.sampler, layout=layout, learnable_volume=True)
        st.dirty_box = ((0, n), (0, n), (n // 2, n // 2 + 1))
        st.capture()
        leg[f"{layout}_one_slice_ms_per_step"] = time_events(st.replay, 30)["median"]
        del st
    leg["note"] = ("headline workload with the volume re-converted inside every step (paired: diffus_pair_volume, bricked: "
                   "diffus_brick_volume, canonical: no conversion, kernels read the caller's tensor); *_one_slice: the caller "
                   "rewrites one dim-2 slice per step and sets CapturedStep.dirty_box (diffus_convert_volume_box)")
    out["learnable_volume"] = leg
    # (c) poses that move: a different ring of poses every step, copied in place into the captured step's buffers
    pool = [pose_ring(n, args.poses, args.rays, phase=0.013 * i) for i in range(16)]
    pool = [(torch.from_numpy(s).to(dev), torch.from_numpy(d).to(dev)) for s, d in pool]
    # (fans="planar": the rings are the reference's fans -- dim-2 component 0, src/cone.py:258 --, and with "auto" the first
    # set_poses() would make that unknown and switch the scatter to its slab-capable launch)
    st = CapturedStep(vol, pool[0][0].clone(), pool[0][1].clone(), args.samples, args.alpha, args.sampler, layout=args.layout,
                      fans="planar")
    st.capture()
    k = [0]

    def moving():
        s, d = pool[k[0] % len(pool)]
        k[0] += 1
        st.set_poses(s, d)
        st.replay()

    ms_move = time_events(moving, 48)["median"]
    ms_fixed = time_events(st.replay, 48)["median"]
    # (d) the reference's training notebook (`[DEMO] Train MRI to Impedance MLP - GPU` cell 16) as whole captured iterations:
    #     MLP -> slice -> frame -> loss -> backward -> Adam with the MSE fused into the renderer, and the cell's real chain
    #     (rotate_around_apex -> differentiable_splat -> min-max -> 1 - SSIM); examples/train_*.py
    try:
        sys.path.insert(0, os.path.join(ROOT, "examples"))
        from train_impedance_mlp import Loop
        from train_ssim_chain import SsimLoop
        loops = {}
        for name, mk in (("mlp_mse", Loop), ("mlp_splat_ssim", SsimLoop)):
            lp = mk()
            lp.iteration()
            lp.capture()
            one = lp.run(200)
            lp.capture(repeat=8)
            loops[name] = {"ms_per_iteration_one_graph_each": one, "ms_per_iteration_8_per_graph": lp.run(200),
                           "final_loss": float(lp.loss)}
            del lp
        loops["note"] = ("64 rays x 228 samples, start 110, 256^3 volume, nearest sampling (the reference's), Adam; one hipGraph per "
                         "iteration, and 8 iterations per graph (two graph launches are ~8.6 us apart on this stack)")
        out["training_loops"] = loops
    except Exception as e:
        out["training_loops"] = {"failed": repr(e)}
    del st
    # (e) fans that are NOT planar in dim 2 -- what a probe-pose optimisation produces (`plot_beam_frame` takes any
    #     `directions`, src/renderer.py:119-124; notebooks/[NW] alignement.ipynb cells 13-14): the headline's poses with every
    #     fan rolled about its central ray, pitched out of the slice, or lying in another coordinate plane.  Same one-pass step.
    tilt = {}
    for name, kw in (("roll5", dict(roll_deg=5.0)), ("roll20", dict(roll_deg=20.0)), ("roll45", dict(roll_deg=45.0)),
                     ("pitch20", dict(pitch_deg=20.0)), ("plane02", dict(plane=(0, 2)))):
        s_t, d_t = pose_ring(n, args.poses, args.rays, **kw)
        st = CapturedStep(vol, torch.from_numpy(s_t).to(dev), torch.from_numpy(d_t).to(dev), args.samples, args.alpha, args.sampler,
                          layout=args.layout)
        for _ in range(5):
            st.step()
        tilt[name + "_ms_per_step"] = time_events(st.step, 48)["median"]
        tilt[name + "_scatter_ms"] = time_events(lambda: st.bwd(_lib_mod.BWD_SCATTER), 48, pre=st.finish_grad)["median"]
        st.finish_grad()
        del st
    st = CapturedStep(vol, sa, da, args.samples, args.alpha, args.sampler, layout=args.layout, fans="oblique")
    for _ in range(5):
        st.step()
    tilt["planar_fans_on_the_slab_capable_launch_ms_per_step"] = time_events(st.step, 48)["median"]
    del st
    tilt["note"] = ("eager one-pass steps of the headline workload (event-timed; the headline's own figure this way is ~2 us above "
                    "its graph/eager best) with tilted fans: the scatter takes its slab path (height-field tile over the fan's "
                    "plane, csrc/scatter.hip); before round 5 these fans fell to the 3-D brick tile and per-sample global "
                    "atomics: 0.083 / 0.172 / 0.394 / 0.519 / 0.118 ms per step (profiles/r05_tilt_before.txt; after: profiles/r05_tilt_after.txt)")
    out["tilted_fan"] = tilt
    # (e') a six-degree-of-freedom probe registration on ONE full-size frame (config 2's: 256 rays x 512 steps), the loop
    #      `[NW] alignement` cells 13-14 attempt: FanPose (csrc/pose.hip) -> render -> sum of squares -> backward -> Adam, eager
    #      through torch autograd and as one captured graph per iteration (examples/register_probe_pose.py)
    try:
        sys.path.insert(0, os.path.join(ROOT, "examples"))
        from register_probe_pose import run as register_run
        reg = {}
        for name, graph, one_pass in (("eager", False, False), ("graph", True, False), ("one_pass_eager", False, True),
                                      ("one_pass_graph", True, True)):
            stats = {}
            hist, apex_err, ang = register_run(iters=400, n=n, R=args.rays, S=args.samples, alpha=args.alpha, report=399, graph=graph,
                                               quiet=True, stats=stats, one_pass=one_pass)
            reg[name + "_ms_per_iteration"] = stats["ms_per_iteration"]
            reg[name + "_final"] = {"loss": hist[-1][1], "apex_error_voxels": apex_err, "worst_ray_angle_deg": ang}
        stats = {}
        register_run(iters=300, n=n, R=args.rays, S=args.samples, alpha=args.alpha, report=10 ** 9, graph=True, quiet=True, stats=stats,
                     one_pass=True, poses=args.poses)
        reg["sweep_one_pass_graph_ms_per_iteration"] = stats["ms_per_iteration"]
        reg["sweep_poses"] = args.poses
        reg["note"] = ("start 3.0 voxels and 5 degrees (roll + pitch) away from the pose that rendered the target; 400 Adam steps; "
                       "wall time per iteration; one_pass: render + loss + backward as CapturedStep.mse_loss (diffus_render_step_mse, no volume "
                       "gradient); sweep: that many frames (probe positions on a ring) registered together, one FanPose module and one render "
                       "launch per iteration for all; before csrc/pose.hip the eager iteration was 1.56 ms (profiles/r05_registration_host.txt)")
        out["registration_6dof"] = reg
    except Exception as e:
        out["registration_6dof"] = {"failed": repr(e)}
    # (f) a SHARED learnable volume on N ranks: what the gradient's collective moves per step (SURVEY §8e "Collective").  A
    #     one-rank RCCL group on this GPU (the N = 8 job cannot be run here): the dense all_reduce of d/dvolume against
    #     allreduce_box (the one slice the reference's training loop learns) and allreduce_touched (the bricks a step touched)
    try:
        import torch.distributed as dist
        from diffus_amd.distributed import allreduce_box, allreduce_touched, allreduce_volume_grad
        own_group = not dist.is_initialized()
        if own_group:
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1)
        st = CapturedStep(vol, sa, da, args.samples, args.alpha, args.sampler, layout=args.layout, learnable_volume="slice" if args.layout != "canonical" else True)
        k = n // 2
        box = ((0, n), (0, n), (k, k + 1))
        scratch = torch.empty(n * n, dtype=torch.float32, device=dev)
        moved = {}

        def step_dense():
            st.step(); allreduce_volume_grad(st.gvol)

        def step_box():
            st.step(); moved["box"] = allreduce_box(st.gvol, box, scratch=scratch)

        def step_touched():
            st._stamp += 1
            st.zero_grad(); st.step_mse(_lib_mod.BWD_ALL)
            moved["touched"] = allreduce_touched(st.gvol_k, st.touched)
            st.finish_grad()

        for f in (step_dense, step_box, step_touched):
            for _ in range(3):
                f()
        shared = {"dense_all_reduce_ms_per_step": time_wall(step_dense, 30), "allreduce_box_one_slice_ms_per_step": time_wall(step_box, 30),
                  "allreduce_touched_ms_per_step": time_wall(step_touched, 30), "step_alone_ms": time_wall(st.step, 30),
                  "bytes_dense": int(st.gvol.numel() * 4), "bytes_box": int(moved["box"]), "bytes_touched": int(moved["touched"]),
                  "note": "wall ms per step (eager one-pass step + the collective on a ONE-rank RCCL group: what the call costs this "
                          "rank, not the xGMI transfer, which a one-rank group does not make); bytes = what goes through the "
                          "collective per step.  allreduce_touched waits for the host once per step (the size of the union)"}
        out["shared_volume_gradient"] = shared
        del st
        if own_group:
            dist.destroy_process_group()
    except Exception as e:
        out["shared_volume_gradient"] = {"failed": repr(e)}
    out["moving_poses"] = {"ms_per_step": ms_move, "fixed_poses_ms_per_step": ms_fixed,
                           "note": "16 rings of poses, 0.013 rad apart, cycled: every step the persistent gradient tensor meets "
                                   "bricks the previous step wrote and this one does not (stale-brick clearing is exercised)"}
    return out


# ------------------------------------------------------------------------------------------------------------------
# Bucketed loss gather (N > 1): step k writes its P losses into slot k % K of ring (k // K) & 1; a full ring leaves in
# one all_gather of K * P floats per rank.  Shared by the GPU worker and the CPU dry run (which tests exactly this).
RING_DEPTH = 8


def ring_slot(k, K, depth=2):
    return k % K, (k // K) % depth


def losses_of_step(gathered_ring, world, K, P, k):
    """The (world * P,) losses of step k, in pose order, out of the gathered ring that holds it."""
    return gathered_ring.view(world, K, P)[:, k % K, :].reshape(-1)


def dry_run(args, world, rank):
    """Launcher / rendezvous / sharding / gather / timing logic without any GPU work (CPU-only boxes, tests)."""
    import torch
    import torch.distributed as dist
    if rank == args.fail_rank:
        print(f"rank {rank}: failing on request (--fail-rank)", file=sys.stderr)
        sys.exit(3)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo")
    scaling, P_total, P = plan_poses(args, world)
    local = torch.arange(rank * P, (rank + 1) * P, dtype=torch.float32) + 1.0     # stand-in loss of pose p: p + 1
    allv = torch.empty(P_total)
    K = max(1, args.gather_every)
    ring = [torch.zeros((K, P)) for _ in range(2)]
    gathered = [torch.zeros(world * K * P) for _ in range(2)]

    def send(b):
        if world > 1:
            dist.all_gather_into_tensor(gathered[b], ring[b].view(-1))
        else:
            gathered[b].copy_(ring[b].view(-1))

    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        j, b = ring_slot(k, K)
        ring[b][j] = local + 1000.0 * k                       # stand-in for step k's losses
        if j == K - 1:
            send(b)
    if args.steps % K:                                        # a part-filled ring goes out too
        send(ring_slot(args.steps - 1, K)[1])
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if args.steps > 0:
        last = args.steps - 1
        allv.copy_(losses_of_step(gathered[ring_slot(last, K)[1]], world, K, P, last) - 1000.0 * last)
    else:
        allv.copy_(torch.arange(P_total, dtype=torch.float32) + 1.0)
    per_rank = [dt]
    counts = [P]
    if world > 1:
        t = torch.tensor([dt, float(P)], dtype=torch.float64)
        every = torch.empty(2 * world, dtype=torch.float64)
        dist.all_gather_into_tensor(every, t)
        per_rank = [float(x) for x in every[0::2]]
        counts = [int(x) for x in every[1::2]]
        dt = max(per_rank)
    assert torch.equal(allv, torch.arange(P_total, dtype=torch.float32) + 1.0), "loss gather out of order"
    if rank == 0:
        print(json.dumps({"metric": "ray-steps/sec fwd+bwd", "value": None, "unit": "ray-steps/s", "n_gpus": world,
                          "world_size_observed": dist.get_world_size() if world > 1 else 1, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": dt / max(args.steps, 1) * 1e3,
                          "per_rank_ms_per_step": [x / max(args.steps, 1) * 1e3 for x in per_rank],
                          "higher_is_better": True, "scaling": scaling, "poses_total": P_total, "poses_per_rank": counts,
                          "config": {"workload": config_label(args, world, P_total)},
                          "dry_run": True, "data": "none (dry run: no GPU work)"}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


_RAW = {}


def raw_nccl(torch, dist, world):
    if "comm" not in _RAW:
        _RAW["comm"] = RawNccl(torch, dist, dist.get_rank(), world)
    return _RAW["comm"]


class RawNccl:
    """ncclAllGather through RCCL's C entry point on a communicator of the bench's own (the unique id travels through the
    torch.distributed group that already exists).  One ctypes call per collective instead of c10d's Work / event / watchdog
    bookkeeping: ~10 us of host time instead of ~50 -- what keeps a 59 us step with a gather every step device-bound."""

    class _Uid(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    def __init__(self, torch, dist, rank, world):
        # Everything that can fail on ONE rank alone (the library's path, dlopen, a missing symbol, ncclGetUniqueId) happens
        # first and its outcome is agreed on by all ranks; only then do they enter the collectives below.  (a rank
        # that failed here used to fall back to torch.distributed by itself while its peers waited in the broadcast.)
        err, lib, uid = None, None, RawNccl._Uid()
        try:
            path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            self.lib = lib = C.CDLL(path)
            lib.ncclGetErrorString.restype = C.c_char_p
            lib.ncclGetErrorString.argtypes = [C.c_int]
            lib.ncclGetUniqueId.restype = C.c_int
            lib.ncclGetUniqueId.argtypes = [C.POINTER(RawNccl._Uid)]
            lib.ncclCommInitRank.restype = C.c_int
            lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, RawNccl._Uid, C.c_int]
            lib.ncclAllGather.restype = C.c_int
            lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
            lib.ncclCommDestroy.restype = C.c_int
            lib.ncclCommDestroy.argtypes = [C.c_void_p]
            if rank == 0:
                self._check(lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        except Exception as e:      # noqa: BLE001 -- reported below, on every rank
            err = e
        if world > 1:
            ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok) == 0:
                raise RuntimeError(f"raw RCCL communicator unavailable on some rank (this rank: {err!r}); all ranks fall back together")
        elif err is not None:
            raise err
        box = [C.string_at(C.byref(uid), 128) if rank == 0 else None]     # (uid.internal would stop at the first NUL byte)
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        C.memmove(C.byref(uid), box[0], 128)
        self.comm = C.c_void_p()
        self._check(lib.ncclCommInitRank(C.byref(self.comm), world, uid, rank), "ncclCommInitRank")

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self.lib.ncclGetErrorString(rc).decode()}")

    def all_gather(self, send, recv, stream):
        """recv (world * n floats) <- every rank's send (n floats), on the HIP stream `stream` (a torch.cuda.Stream)"""
        self._check(self.lib.ncclAllGather(C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), send.numel(), 7,   # 7 = ncclFloat32
                                           self.comm, C.c_void_p(stream.cuda_stream)), "ncclAllGather")

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()


class StepRunner:
    """One CapturedStep, its ring of per-pose loss buffers (one captured hipGraph per slot) and, with N > 1, the loss
    all_gather on a communication stream.  `timed(steps, warmup)` is the contract's timed region: warm-up, barrier +
    synchronize, EXACTLY `steps` steps, barrier + synchronize, MAX over ranks."""

    def __init__(self, hp, args, dev, dist=None, world=1, K=1, eager=False):
        import torch
        self.torch, self.hp, self.args, self.dev, self.dist, self.world = torch, hp, args, dev, dist, world
        issue = "eager" if (eager or args.eager) else args.issue
        eager = issue == "eager"
        self.issue_probe = None
        P = hp.P
        self.P = P
        self.nccl = dist is not None and args.dist_backend == "nccl"
        self.K = K = 1 if (dist is None or not self.nccl or args.sync_gather) else max(1, K)
        # Ring of loss buffers: a buffer is rewritten `depth` gathers after it left.  Whether its gather has finished is
        # asked on the HOST (event query; a host wait if the host ever gets that far ahead): a wait_event on the compute
        # stream costs ~6 us of device time per step (a barrier packet between the step's kernels and the next step's),
        # measured on a one-rank RCCL group: 0.0743 -> 0.0681 ms per step.
        self.depth = D = max(2, RING_DEPTH // K)
        self.losses_all = torch.empty((P * world,), dtype=torch.float32, device=dev)
        self.ring = [torch.ones((K, P), dtype=torch.float32, device=dev) for _ in range(D)]
        self.graphs = [[None] * K for _ in range(D)]
        self.graph = None
        self.kstep = 0
        side = torch.cuda.Stream()
        if not eager:
            try:
                with torch.cuda.stream(side):
                    for _ in range(3):
                        hp.step()
                side.synchronize()
                for b in range(D):
                    for j in range(K):
                        hp.loss = self.ring[b][j]
                        self.graphs[b][j] = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(self.graphs[b][j], stream=side):
                            hp.step()
                self.graph = self.graphs[0][0]
            except Exception as e:  # capture unsupported -> eager, and say so
                print(f"hipGraph capture failed ({e!r}); running eagerly", file=sys.stderr)
                self.graph = None
        hp.loss = self.ring[0][0]
        if issue == "auto" and self.graph is not None:
            # both ways, a few steps each, on this box, now: the faster one issues the timed steps
            def probe(fn, n=60):
                for _ in range(10):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n * 1e3
            g_ms = probe(self.graphs[0][0].replay)
            e_ms = probe(hp.step)
            self.issue_probe = {"eager_ms_per_step": e_ms, "graph_ms_per_step": g_ms}
            if e_ms < g_ms:
                self.graph = None
        # The one collective of the path: all_gather of the per-pose losses over xGMI, on its own stream (an event per
        # ring orders reuse).  Every gather has completed before the closing barrier of the timed region.
        self.overlap = self.nccl and not args.sync_gather
        self.raw = None
        if self.overlap and args.raw_nccl_gather:
            try:
                self.raw = raw_nccl(torch, dist, world)
            except Exception as e:
                print(f"raw ncclAllGather unavailable ({e!r}); gathering through torch.distributed", file=sys.stderr)
        if self.overlap:
            try:
                self.comm = torch.cuda.Stream()
                self.gathered = [torch.ones((world * K * P,), dtype=torch.float32, device=dev) for _ in range(D)]
                self.full_ev = [torch.cuda.Event() for _ in range(D)]
                self.gather_ev = [torch.cuda.Event() for _ in range(D)]
            except Exception as e:
                print(f"overlapped gather unavailable ({e!r}); gathering on the compute stream", file=sys.stderr)
                self.overlap = False

    def send(self, b):                                 # ring b -> every rank, on the communication stream
        torch = self.torch
        self.full_ev[b].record(torch.cuda.current_stream())
        if self.raw is not None:
            self.comm.wait_event(self.full_ev[b])
            self.raw.all_gather(self.ring[b], self.gathered[b], self.comm)
            self.gather_ev[b].record(self.comm)
            return
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(self.full_ev[b])
            self.dist.all_gather_into_tensor(self.gathered[b], self.ring[b].view(-1))
            self.gather_ev[b].record(self.comm)

    def step(self):
        torch, K = self.torch, self.K
        k = self.kstep
        j, b = ring_slot(k, K, self.depth) if self.overlap else (0, 0)
        if self.overlap and j == 0 and k >= self.depth * K and not self.gather_ev[b].query():
            self.gather_ev[b].synchronize()                     # ring b is not free yet: the host waits, not the stream
        if self.graph is not None:
            self.graphs[b][j].replay()
        else:
            self.hp.loss = self.ring[b][j]
            self.hp.step()
        self.kstep = k + 1
        if self.overlap:
            if j == K - 1:
                self.send(b)
        elif self.nccl:
            self.dist.all_gather_into_tensor(self.losses_all, self.ring[0][0])
        elif self.dist is not None:                             # gloo rehearsal: through host memory
            out = torch.empty(self.P * self.world, dtype=torch.float32)
            self.dist.all_gather_into_tensor(out, self.ring[0][0].cpu())
            self.losses_all.copy_(out)

    def barrier(self):
        torch = self.torch
        if self.overlap:
            if self.kstep % self.K:                             # a part-filled ring goes out too
                self.send(ring_slot(self.kstep - 1, self.K, self.depth)[1])
            torch.cuda.current_stream().wait_stream(self.comm)  # every gather issued so far is part of the step count
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def last_slot(self):
        j, b = ring_slot(self.kstep - 1, self.K, self.depth) if self.overlap else (0, 0)
        return self.ring[b][j]

    _frozen = [False]

    @staticmethod
    def _gc_quiet():
        """Collect now; the first time also freeze what is alive (torch, numpy: ~1e6 objects), so that later collections
        only walk what was allocated since and take microseconds instead of ~30 ms of idle GPU."""
        import gc
        gc.collect()
        if not StepRunner._frozen[0]:
            gc.freeze()
            StepRunner._frozen[0] = True

    def prewarm(self, ms):
        """Untimed steps for ~`ms` milliseconds BEFORE the W warm-up steps of timed(): the chip's clocks settle over tens of
        milliseconds of load (measured: the 20-step region the driver times reads 0.0566-0.0602 ms per step after 125 steps of
        run-up and 0.0554 after 420), and a 1.1 ms timed region is all run-up otherwise."""
        torch = self.torch
        StepRunner._gc_quiet()      # (here, not between the run-up and the timed region: see timed())
        self._prewarmed = True
        if self.dist is None:
            t_end = time.perf_counter() + ms * 1e-3
            n = 0
            while time.perf_counter() < t_end:
                for _ in range(16):
                    self.step()
                torch.cuda.synchronize()
                n += 16
            return n
        # N > 1: every step is (or feeds) a collective, so every rank must run the SAME number of steps: each rank times 16 of
        # them, the ranks agree on the largest count any of them proposes (one all_reduce), and all run exactly that many
        t0 = time.perf_counter()
        for _ in range(16):
            self.step()
        torch.cuda.synchronize()
        per = max((time.perf_counter() - t0) / 16, 1e-6)
        want = torch.tensor([min(max(int(ms * 1e-3 / per), 16), 20000)], dtype=torch.int64, device=self.dev if self.nccl else "cpu")
        self.dist.all_reduce(want, op=self.dist.ReduceOp.MAX)
        n = (int(want.item()) + 15) // 16 * 16
        for _ in range(n // 16):
            for _ in range(16):
                self.step()
            torch.cuda.synchronize()
        return n + 16

    def timed(self, steps, warmup):
        """-> dict(dt = seconds of the timed region, MAX over ranks; per_rank_ms; host_ms; world_seen)."""
        import gc
        torch, dist = self.torch, self.dist
        # No cyclic garbage collection inside the timed region (what `timeit` does too): a generation-2 pass over the ~1e6
        # objects torch and numpy leave alive is ~10 ms -- ten times a 20-step region (seen once in four such runs).  Collected
        # BEFORE the warm-up steps: the collection itself idles the GPU for tens of ms, and a chip that has idled that long
        # starts the timed region at low clocks (0.063 ms per step instead of 0.055).
        gc_was = gc.isenabled()
        if not getattr(self, "_prewarmed", False):      # (the secondary legs: a short run-up of their own; the headline has had prewarm())
            self.prewarm(20.0)
        self._prewarmed = False                         # one timed region per run-up
        gc.disable()
        for _ in range(warmup):
            self.step()
        self.barrier()
        try:
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            host_ms = (time.perf_counter() - t0) * 1e3 / steps    # host time to ENQUEUE one step (no device wait)
            self.barrier()
            dt = time.perf_counter() - t0
        finally:
            if gc_was:
                gc.enable()
        per_rank_ms = [dt / steps * 1e3]
        world_seen = 1
        if dist is not None:
            on = self.dev if self.nccl else "cpu"
            mine = torch.tensor([dt], dtype=torch.float64, device=on)
            every = torch.empty(self.world, dtype=torch.float64, device=on)
            dist.all_gather_into_tensor(every, mine)
            per_rank_ms = [float(x) / steps * 1e3 for x in every.cpu()]
            dt = float(every.max().item())                      # MAX over ranks
            world_seen = dist.get_world_size()
            if self.overlap:                                    # the last step's slot of every rank, in pose order
                last = self.kstep - 1
                self.losses_all.copy_(losses_of_step(self.gathered[ring_slot(last, self.K, self.depth)[1]], self.world, self.K, self.P, last))
        else:
            self.losses_all.copy_(self.last_slot())
        return {"dt": dt, "per_rank_ms": per_rank_ms, "host_ms": host_ms, "world_seen": world_seen}


def verify_step(args, vol_of_pose, src_all, dirs_all, local_lo, hp, losses_all, frame_poses, loss_poses):
    """The timed step checks what it computed, OUTSIDE the timed region: frames of `frame_poses`
    (indices into this rank's shard) and per-pose losses of `loss_poses` (global pose indices, read from the GATHERED
    vector, so a gather that scrambles the pose order fails here) against the CPU oracle (oracle/diffus_oracle.c: the
    restatement of reference src/renderer.py:201-275; echo series in float64).  Frame tolerance: 2e-5 max-norm-relative,
    widened on fans that graze the skull to 10 input roundings' worth (oracle/conditioning.py -- the reference's own
    float32 LU is 3e-5 .. 1.4e-4 from its float64 result on such rays, golden G17).  loss_p = sum(frame_p^2): 1e-4.
    `vol_of_pose(p)` = the volume the OWNER of global pose p rendered (config 5: one phantom variant per rank)."""
    import numpy as np
    from oracle import oracle as orc
    from oracle.conditioning import frame64_and_tolerance
    orc.build()
    cache = {}

    def oracle_frame(p):
        if p not in cache:
            if args.start == 0:
                f64, tol, _ = frame64_and_tolerance(vol_of_pose(p), src_all[p], dirs_all[p], args.samples, args.alpha, sampler=args.sampler)
            else:       # start crop + median: the float32 scalar oracle, fixed tolerance
                f64 = orc.plot_beam_frame(vol_of_pose(p), src_all[p], dirs_all[p], args.samples, args.alpha, args.start,
                                          sampler=args.sampler)[3].astype(np.float64)
                tol = 1e-4
            cache[p] = (f64, tol)
        return cache[p]

    frames, losses, ok = [], [], True
    for q in frame_poses:
        f64, tol = oracle_frame(local_lo + q)
        err = float(np.abs(hp.frame[q].cpu().numpy() - f64).max() / max(np.abs(f64).max(), 1e-300))
        frames.append({"pose": int(local_lo + q), "rel_err": err, "tol": tol})
        ok &= err <= tol
    # the same poses through the FORWARD kernel (diffus_render_fwd: what plot_beam_frame / render_poses return).  It evaluates
    # ill-conditioned rays (|echo| > 1: fans that graze the skull) again in float64 inside the kernel; the one-pass training step
    # timed above does not unless asked to (DIFFUS_BWD_REPAIR_FRAME, ~10 us per step) -- its frame is a by-product.  Tolerance:
    # 5e-5 (the reference's own float32 result is 4.2e-5 from this on pose 18, golden G19).
    fwd_frames = []
    if args.start == 0 and frame_poses:
        hp.fwd()
        for q in frame_poses:
            f64, _ = oracle_frame(local_lo + q)
            err = float(np.abs(hp.frame[q].cpu().numpy() - f64).max() / max(np.abs(f64).max(), 1e-300))
            fwd_frames.append({"pose": int(local_lo + q), "rel_err": err, "tol": 5e-5})
            ok &= err <= 5e-5
        hp.step()       # leave the step's own frame in the buffer again
    lv = losses_all.cpu().numpy()
    for p in loss_poses:
        want = float((oracle_frame(p)[0] ** 2).sum())
        err = abs(float(lv[p]) - want) / max(abs(want), 1e-300)
        losses.append({"pose": int(p), "rel_err": err, "tol": 1e-4})
        ok &= err <= 1e-4
    return {"oracle": "oracle/diffus_oracle.c + oracle/conditioning.py (CPU restatement of reference src/renderer.py:201-275, "
                      "echo series in float64)",
            "frames": frames, "forward_kernel_frames": fwd_frames, "losses": losses,
            "max_rel_err": max([x["rel_err"] for x in frames + losses] or [0.0]), "ok": bool(ok)}


def worker(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        return dry_run(args, world, rank)

    # The contract is ONE JSON line on stdout.  RCCL (and anything else underneath) writes its warnings to file
    # descriptor 1: from here on fd 1 is stderr, and rank 0 writes the line to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from diffus_amd import CapturedStep, _lib
    from diffus_amd.phantom import phantom, pose_ring

    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29533"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl":
            if local >= torch.cuda.device_count():              # the launcher does not touch HIP: every rank checks itself
                print(f"bench.py: rank {rank}: LOCAL_RANK {local} but {torch.cuda.device_count()} GPU(s) visible", file=sys.stderr)
                sys.exit(2)
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; reporting n_gpus = {world}", file=sys.stderr)
    ngpu = world
    dev = torch.device("cuda", torch.cuda.current_device())

    scaling, P_total, P = plan_poses(args, world)
    # config 5 (--n 512): one distinct volume per GPU -- the phantom variant (tumour position) is the rank
    vol_np = phantom(args.n, variant=rank if args.n >= 512 else 0)
    vol = torch.from_numpy(vol_np).to(dev)

    def shard(total, per_rank):
        s_all, d_all = pose_ring(args.n, total, args.rays)
        lo = rank * per_rank
        return (s_all, d_all, lo, torch.from_numpy(s_all[lo:lo + per_rank]).to(dev).contiguous(),
                torch.from_numpy(d_all[lo:lo + per_rank]).to(dev).contiguous())

    src_all, dirs_all, lo, src, dirs = shard(P_total, P)

    def make_step(s, d, want_gvol=not args.no_gvol, learnable=args.learnable_volume):
        return CapturedStep(vol, s, d, args.samples, args.alpha, args.sampler, start=args.start, want_gvol=want_gvol,
                            layout=args.layout, sparse=not args.dense_grad, persistent=not args.memset_grad,
                            learnable_volume=learnable, fused_loss=not args.unfused_loss, one_pass=not args.two_pass)

    # --- the headline: eager launches, or one captured hipGraph per loss slot (compute) + the collective ---
    # N > 1: the per-pose losses leave in one all_gather per step on the communication stream (ring of K = 1 slot, twice:
    # the gather of step k reads its buffer while step k+1 already writes the other one).  --gather-every K buckets K
    # steps into one collective; that variant is timed as well and reported beside the headline.
    hp = make_step(src, dirs)
    run = StepRunner(hp, args, dev, dist, world, K=args.gather_every, eager=args.eager)
    prewarm_steps = run.prewarm(args.prewarm_ms) if args.prewarm_ms > 0 else 0
    res = run.timed(args.steps, args.warmup)
    dt = res["dt"]
    ray_steps = P_total * args.rays * args.samples
    value = ray_steps * args.steps / dt
    if dist is not None:      # every rank must hold all P losses, in pose order, and they must be finite
        assert torch.isfinite(run.losses_all).all() and float(run.losses_all.abs().min()) > 0, "loss gather failed"

    # --- the step checks itself (outside the timed region): rank 0's first and last pose as frames, and the gathered
    # losses of those two plus the LAST pose of the whole job (the last rank's, so the gather's pose order is covered) ---
    verified = None
    if rank == 0 and not args.no_verify:
        try:
            loss_poses = sorted({0, P - 1, P_total - 1})
            vols = {0: vol_np}                        # config 5: pose p was rendered through its owner's phantom variant

            def vol_of_pose(p):
                v = (p // P) if args.n >= 512 else 0
                if v not in vols:
                    for k in [k for k in vols if k != 0]:
                        del vols[k]                   # 512 MiB each: rank 0's and one other at a time
                    vols[v] = phantom(args.n, variant=v)
                return vols[v]
            # (P * 9) // 16: pose 18 of config 3, a fan that grazes the skull -- the hard case is checked too
            verified = verify_step(args, vol_of_pose, src_all, dirs_all, lo, hp, run.losses_all,
                                   sorted({0, (P * 9) // 16, P - 1}), loss_poses)
        except Exception as e:      # the oracle could not be built or run: say so, never claim a check that did not happen
            verified = {"ok": False, "failed": repr(e)}

    # --- a thicker window: the same step 200 more times (the driver's own K may be as small as 20) ---
    extra = {}
    if dist is None and args.steps < 200:
        r200 = run.timed(200, 0)
        extra["steps_200"] = {"ms_per_step": r200["dt"] / 200 * 1e3, "value": ray_steps * 200 / r200["dt"]}

    # --- the other leg of the scaling story, measured in the same run ---
    #   strong (N > 1): the weak figure (--poses per GPU) and the one-GPU leg of the strong curve (all P_total poses on
    #                   rank 0's GPU alone, no collective) -> speedup_vs_one_gpu
    #   N = 1, auto:    BASELINE config 4 at 1 of 8 GPUs (P_total poses on this GPU) = the base of the strong curve
    scale_info = {}
    if not args.no_scaling_legs:
        def one_gpu_leg(total):
            s_all, d_all = pose_ring(args.n, total, args.rays)
            h = make_step(torch.from_numpy(s_all).to(dev).contiguous(), torch.from_numpy(d_all).to(dev).contiguous())
            r = StepRunner(h, args, dev, None, 1, eager=args.eager).timed(args.steps, args.warmup)
            return {"poses_total": total, "n_gpus": 1, "ms_per_step": r["dt"] / args.steps * 1e3,
                    "value": total * args.rays * args.samples * args.steps / r["dt"],
                    "workload": config_label(args, 1, total)}
        if scaling == "strong" and world > 1:
            if P != args.poses:
                _, _, _, s_w, d_w = shard(args.poses * world, args.poses)
                rw = StepRunner(make_step(s_w, d_w), args, dev, dist, world, K=args.gather_every, eager=args.eager)
                w = rw.timed(args.steps, args.warmup)
                scale_info["weak"] = {"poses_per_gpu": args.poses, "poses_total": args.poses * world,
                                      "ms_per_step": w["dt"] / args.steps * 1e3,
                                      "value": args.poses * world * args.rays * args.samples * args.steps / w["dt"]}
                del rw
            else:
                scale_info["weak"] = {"poses_per_gpu": P, "poses_total": P_total, "ms_per_step": dt / args.steps * 1e3,
                                      "value": value, "note": "identical to the headline at this N"}
            if rank == 0:
                base = one_gpu_leg(P_total)
                scale_info["strong"] = {"poses_total": P_total, "poses_per_gpu": P, "one_gpu": base,
                                        "speedup_vs_one_gpu": value / base["value"]}
            dist.barrier()
        elif world == 1 and args.scaling == "auto" and config_label(args, 1, P_total) == "BASELINE config 3":
            scale_info["strong_base"] = one_gpu_leg(args.poses_total)
            # the value a reader of the N = 1, 2, 4, 8 lines must divide the N > 1 (strong, config 4) values by
            scale_info["scale_base_value"] = scale_info["strong_base"]["value"]
            scale_info["note"] = ("`value` is BASELINE config 3 (32 poses on one GPU).  With --gpus N > 1 this script shards "
                                  f"config 4's {args.poses_total} poses over the N ranks (strong scaling); the one-GPU leg of THAT "
                                  "curve is strong_base.value, not `value`")
        if args.gather_every == 1 and dist is not None and run.nccl and not args.sync_gather:
            rb = StepRunner(make_step(src, dirs), args, dev, dist, world, K=8, eager=args.eager)
            bk = rb.timed(args.steps, args.warmup)
            scale_info["bucketed_gather"] = {"gather_every": 8, "ms_per_step": bk["dt"] / args.steps * 1e3,
                                             "value": ray_steps * args.steps / bk["dt"],
                                             "note": "opt-in (--gather-every 8): the losses of 8 steps leave in one all_gather"}
            del rb

    # --- per-kernel device time (HIP events on the launch stream), rank-local ---
    it = max(10, min(args.steps, 50))
    hp.fwd(); hp.loss_and_grad()
    one_pass = hp.fused_loss and hp.one_pass
    if one_pass:    # the step has no forward launch: its scan kernel writes the frame too
        k_ms = {"render_bwd_kernel": time_events(lambda: hp.step_mse(_lib.BWD_SCAN, epilogue=False), it)}
    else:
        k_ms = {"render_fwd_kernel": time_events(hp.fwd, it),
                "render_bwd_kernel": time_events((lambda: hp.bwd_mse(_lib.BWD_SCAN)) if hp.fused_loss else
                                                 (lambda: hp.bwd(_lib.BWD_SCAN)), it)}
    if not args.no_gvol:
        k_ms["scatter_patch_kernel"] = time_events(lambda: hp.bwd(_lib.BWD_SCATTER), it,
                                                   pre=(hp.finish_grad if hp.touched is not None else hp.zero_grad))
        hp.finish_grad()
    if args.start > 0:
        k_ms["note"] = "start > 0: the forward and scan figures include the per-pose median launch"
    local_rs = P * args.rays * args.samples
    b = BYTES[args.sampler]
    kern = {k: v for k, v in k_ms.items() if isinstance(v, dict)}
    dom = max(kern, key=lambda k: kern[k]["mean"])
    dom_ms = kern[dom]["mean"]
    achieved = b[dom] * local_rs / (dom_ms * 1e-3) / 1e9
    # HBM-side bytes per launch from the committed PMC passes, only if they were collected on exactly this workload
    traffic = measured = limiter = pmc_file = None
    evidence = issue = None
    found = find_pmc_summary(workload_key(args, P))
    if found is not None:
        pmc_file, pm = found
        e = pm.get("kernels", {}).get(dom, {})
        traffic = e.get("hbm_bytes_per_launch")
        limiter = e.get("limiter")
        evidence = {k: e[k] for k in ("l2_hit_rate", "valu_busy_frac", "wait_any_frac_of_wave", "active_valu_frac_of_wave",
                                      "lds_busy_frac", "insts_valu_per_wave", "insts_lds_per_wave", "atomic_GBs", "vgpr",
                                      "avg_us") if k in e}
        if "step_valu_wave_insts" in pm:
            evidence["step_valu_wave_insts"] = pm["step_valu_wave_insts"]
        # What the kernel is actually held to (the counters say "valu-issue", not "hbm"): its VALU instruction stream
        # against the SIMDs' issue rate.  Two peaks, both from evidence under profiles/ (tools/valu_issue_bench.hip, round 4):
        #   * the guide's FP32 rate, 2 cycles per wave64 instruction (MI355X_MICROARCH.md "Wave scheduling"; measured 2.2-2.5
        #     for v_fma/v_mul/v_add_f32, v_mov, v_and/v_or, v_add_u32 with >= 2 waves per SIMD);
        #   * cost-weighted: DPP moves, selects, compares, conversions, v_ldexp, min/max, 24-bit multiplies and the
        #     3-operand integer forms take 4.25 cycles, v_rcp/v_exp 8.2 -- the kernel's own mix (tools/issue_model.py on its
        #     disassembly, profiles/r04_issue_model.json) gives its mean cycles per instruction.
        vi = e.get("counters", {}).get("SQ_INSTS_VALU")
        if vi:
            n_simd, clk = 1024, 2.4e9
            peak = n_simd * clk / 2.0
            issue = {"bound": "valu-issue", "wave_insts_per_launch": vi, "achieved": vi / (dom_ms * 1e-3), "peak": peak,
                     "unit": "VALU wave-instructions/s", "frac": vi / (dom_ms * 1e-3) / peak,
                     "note": "SQ_INSTS_VALU of the committed PMC pass over the live launch time; peak = 1024 SIMDs x 2.4 GHz / 2 "
                             "cycles (the FP32 rate of MI355X_MICROARCH.md, confirmed by profiles/r04_valu_issue_bench.txt)"}
            mix = issue_mix(dom)
            if mix:
                cyc = vi * mix["mean_cycles_per_valu"] / n_simd
                issue["cost_weighted"] = {
                    "mean_cycles_per_inst": mix["mean_cycles_per_valu"], "issue_cycles_per_simd": cyc,
                    "frac": cyc / (dom_ms * 1e-3 * clk), "source": mix["source"],
                    "note": "the kernel's instruction mix priced with the MEASURED per-instruction issue costs: the share of the "
                            "launch its SIMDs need just to issue its VALU instructions"}
        else:
            issue = None
        if traffic is not None:
            measured = traffic / (dom_ms * 1e-3) / 1e9
    # the same figures for every timed kernel of the step (scan and scatter are within a few per cent of each other: which
    # of them is "dominant" changes from run to run)
    per_kernel = {}
    for k, t in kern.items():
        rec = {"launch_ms": t["mean"], "algorithmic_GBs": b[k] * local_rs / (t["mean"] * 1e-3) / 1e9}
        if found is not None:
            ek = found[1].get("kernels", {}).get(k, {})
            if ek.get("hbm_bytes_per_launch") is not None:
                rec["traffic"] = ek["hbm_bytes_per_launch"]
                rec["measured_hbm_GBs"] = ek["hbm_bytes_per_launch"] / (t["mean"] * 1e-3) / 1e9
                rec["measured_hbm_frac"] = rec["measured_hbm_GBs"] / HBM_PEAK_GBS
            vk = ek.get("counters", {}).get("SQ_INSTS_VALU")
            mk = issue_mix(k)
            if vk:
                rec["valu_issue_frac_at_fp32_rate"] = vk * 2.0 / 1024 / (t["mean"] * 1e-3 * 2.4e9)
                if mk:
                    rec["valu_issue_frac_cost_weighted"] = vk * mk["mean_cycles_per_valu"] / 1024 / (t["mean"] * 1e-3 * 2.4e9)
            if ek.get("limiter"):
                rec["bound"] = ek["limiter"]
        per_kernel[k] = rec
    if measured is not None and measured / HBM_PEAK_GBS >= 0.4:
        bound = "hbm"
    elif limiter:
        bound = limiter
    elif measured is not None:
        bound = "latency/issue (measured HBM-side traffic is %.0f %% of peak; no SQ counters committed for this workload)" % (100 * measured / HBM_PEAK_GBS)
    else:
        bound = "unmeasured (no PMC summary committed for this workload; the algorithmic model alone cannot name a bound)"

    # --- single-pose latency (BASELINE config 2): 1 pose, fwd + bwd, eager and graph-replayed ---
    hp1 = make_step(src[:1].contiguous(), dirs[:1].contiguous(), learnable=False)
    for _ in range(5):
        hp1.step()
    sp = time_events(hp1.step, 20)
    sp_graph = None
    try:
        g1 = hp1.capture()
        sp_graph = time_events(g1.replay, 20)
    except Exception:
        pass
    # the same frame with the pose-gradient-only backward that config 2 names (no d/dvolume: no scatter, no flush)
    hp1p = make_step(src[:1].contiguous(), dirs[:1].contiguous(), want_gvol=False, learnable=False)
    for _ in range(5):
        hp1p.step()
    sp_pose = time_events(hp1p.step, 20)

    ok = True
    if rank == 0:
        grads = "d/dsource, d/ddirections" if args.no_gvol else "d/dvolume, d/dsource, d/ddirections"
        K, overlap = run.K, run.overlap
        out = {
            "metric": "ray-steps/sec fwd+bwd",
            "value": value,
            "unit": "ray-steps/s",
            "n_gpus": ngpu,
            "world_size_observed": res["world_seen"],
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "per_rank_ms_per_step": res["per_rank_ms"],
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"{config_label(args, ngpu, P_total)}: {P_total} poses ({P} per GPU) x "
                             f"{args.rays} rays x {args.samples} steps through a {args.n}^3 analytic head phantom; "
                             f"{args.sampler} sampling; forward + sum-of-squares loss + backward ({grads}) "
                             f"+ canonical gradient + per-pose loss gather"),
                "loss": ("separate kernel (diffus_loss_sumsq)" if args.unfused_loss else
                         "fused: dL/dframe formed inside the backward, per-pose loss summed by its per-pose blocks"),
                "passes": ("one (diffus_render_step_mse: the adjoint-scan kernel also writes the frame; no forward launch)"
                           if one_pass else "two (diffus_render_fwd, then the backward)"),
                "poses_per_gpu": P, "poses_total": P_total, "rays": args.rays, "samples": args.samples,
                "volume": [args.n] * 3, "sampler": args.sampler, "start": args.start, "alpha": args.alpha,
                "layout": args.layout,
                "volume_conversion": ("inside every step (learnable volume)" if args.learnable_volume else
                                      "once, outside the timed region (constant volume; see callers.learnable_volume)") if args.layout != "canonical" else "none",
                "grad_handback": "dense" if args.dense_grad else ("sparse (touched bricks), persistent tensor" if hp.persistent else "sparse (touched bricks), memset per step"), "issue": ("one captured hipGraph replayed per step" if run.graph is not None else
                                   "eager: the step's launches issued from Python through the C-ABI (two calls, three kernels)"),
                "issue_probe": run.issue_probe,
                "prewarm_steps": prewarm_steps,
                "parallelism": f"poses sharded x{ngpu}, volume replicated" if args.n < 512 else f"one volume per GPU x{ngpu} (replicas only)",
                "host_enqueue_ms_per_step": res["host_ms"],
                "loss_gather": ("none (1 GPU)" if dist is None else
                                (f"one all_gather per {K} step(s) ({K} x P losses per rank) on its own stream, overlapping the next step(s)"
                                 if overlap else "all_gather every step on the compute stream")),
                "dist_backend": None if dist is None else args.dist_backend,
                "gather_call": None if dist is None else ("ncclAllGather (ctypes, the bench's own communicator)" if run.raw is not None
                                                          else "torch.distributed.all_gather_into_tensor"),
            },
            "verified": verified,
            "roofline": {
                # measured first: `achieved` / `frac` are the HBM-side bytes of the committed PMC passes over the
                # live launch time; the contract's no-reuse ALGORITHMIC figure sits beside them under `algorithmic`
                "bound": bound,
                "kernel": dom,
                "achieved": measured,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": None if measured is None else measured / HBM_PEAK_GBS,
                "frac_kind": "measured_hbm_traffic",   # counters over event time; the contract's model figure is `algorithmic.frac_of_peak`
                "traffic": traffic,
                "pmc_summary": pmc_file,
                "algorithmic": {"GBs": achieved, "frac_of_peak": achieved / HBM_PEAK_GBS, "bytes_per_ray_step": b[dom],
                                "ray_steps_per_launch": local_rs,
                                "note": f"no-reuse model, {b[dom]} B per ray-step (SURVEY 8d) x ray-steps per launch / launch time: NOT "
                                        "bytes that crossed the HBM interface -- a planar fan re-reads a cache-resident sheet "
                                        "(traffic / algorithmic = %s)" % ("n/a" if traffic is None else "%.2f" % (traffic / (b[dom] * local_rs)))},
                "issue_roofline": issue,
                "per_kernel": per_kernel,
                "evidence": evidence,
                "launch_ms": dom_ms,
                "kernels_ms": k_ms,
                "whole_step_algorithmic_GBs": (sum(b[k] for k in kern) * local_rs) / (dt / args.steps) / 1e9,
            },
            "single_pose": {"workload": f"{'BASELINE config 2: ' if (args.n, args.rays, args.samples, args.start) == (256, 256, 512, 0) else ''}"
                                        f"1 pose x {args.rays} rays x {args.samples} steps, {args.n}^3 volume, fwd+bwd",
                            "eager_ms": sp["median"], "graph_ms": sp_graph["median"] if sp_graph else None,
                            "pose_gradient_only_ms": sp_pose["median"],
                            "value": args.rays * args.samples / (min(sp["median"], (sp_graph or sp)["median"]) * 1e-3),
                            "value_pose_gradient_only": args.rays * args.samples / (sp_pose["median"] * 1e-3),
                            "note": "`value`: forward + backward to volume, source and directions (three launches); "
                                    "`value_pose_gradient_only`: forward + pose-gradient backward, what BASELINE config 2 names (two "
                                    "launches: scan + per-pose epilogue).  Launch-bound: an ordinary launch is 3.3 us in a stream, a "
                                    "cooperative one 21.7 us and a grid sync ~28 us (profiles/r05_grid_sync_bench.txt)"},
        }
        out.update(extra)
        out.update(scale_info)
        if ngpu == 1 and not args.no_callers:
            try:
                out["callers"] = callers_legs(args, vol, dev)
            except Exception as e:
                out["callers"] = {"failed": repr(e)}
        if ngpu == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the baseline is informative; never lose the GPU line over it
                out["cpu_baseline"] = {"value": None, "unit": "ray-steps/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e!r}"}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
        if verified is not None and not verified.get("ok", False):
            print(f"bench.py: the timed step does NOT match the oracle: {verified}", file=sys.stderr)
            ok = False
    os.close(json_fd)
    if dist is not None:
        torch.cuda.synchronize()
        if "comm" in _RAW:
            _RAW["comm"].close()
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit(4)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not args.dry_run and args.dist_backend == "nccl":
        pin_to_gpu_numa_node(int(os.environ.get("LOCAL_RANK", "0")))      # (each rank, under our launcher or torch.distributed.run)
    worker(args)


if __name__ == "__main__":
    main()
