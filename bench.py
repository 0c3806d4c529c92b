#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the fused rate-control hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    N > 1: one rank per GPU.  Started by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the
    ranks read RANK / LOCAL_RANK / WORLD_SIZE; started plainly (`python bench.py --gpus N`, WORLD_SIZE unset) the parent
    makes NO GPU call: it starts that launcher as a child process and relays rank 0's JSON line.  Env batches shard with
    no data-path collective; the N > 1 line adds a short `train` leg whose flat-gradient all-reduce runs on RCCL.

One "step" = one pass of the hot path over one batch: ONE fused launch of `rate_env_step` over 65 536 envs per GPU
(action clip -> 20 RK4 sub-steps of the 6-DOF model -> command update -> reward -> termination -> observation ->
episode-done compaction -> in-kernel auto-reset), with synthetic actions already resident in HBM.  The default
precision is "mixed" (fp32 derivative evaluations, fp64 state accumulate): the cheapest variant that meets the
north-star 1e-4 parity gate over 1000 steps (tests/test_gpu_parity.py prints the measured drift of each variant).

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs between ranks on this driver stack



def _self_launch_if_needed():
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start N ranks as a CHILD
    `torch.distributed.run` and relay rank 0's line.  Runs before torch is imported, so this process never touches the GPU
    (a process that has initialised the GPU must not exec or re-exec; children are fresh processes)."""
    if "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    n = 1
    argv = sys.argv[1:]
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1:
        return
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE)          # stderr passes through; stdout carries rank 0's JSON line
    line = None
    for ln in proc.stdout.decode(errors="replace").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    sys.exit(proc.returncode if proc.returncode or line is not None else 1)


if __name__ == "__main__":
    _self_launch_if_needed()

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable with a float4 copy)
FP32_VECTOR_PEAK_TFLOPS = 157.3
# SURVEY §8(d) algorithmic bytes per unit (fp32 words) and flops per unit
ALG_BYTES = {"env": 300.0, "env_pid": 300.0, "physics": 112.0, "cascade": 344.0, "rollout": 300.0, "train": 300.0}
ALG_FLOPS = {"env": 40.0e3, "env_pid": 40.0e3 + 0.05e3, "physics": 2.0e3, "cascade": 2.4e3, "rollout": 40.0e3, "train": 40.0e3}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="env", choices=["env", "env_pid", "physics", "cascade", "rollout", "train"])
    ap.add_argument("--precision", default="mixed", choices=["f64", "mixed", "f32"])
    ap.add_argument("--batch", type=int, default=65536, help="envs / aircraft per GPU")
    ap.add_argument("--graph", type=int, default=1, help="replay the step loop from a hipGraph (1) or launch eagerly (0)")
    ap.add_argument("--graph-steps", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the short secondary measurements of the default run")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--policy-dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: CPU-side rendezvous for rehearsals")
    ap.add_argument("--device-index", type=int, default=-1, help="force every rank onto this GPU (1-GPU rehearsal of N>1)")
    ap.add_argument("--single-rank-group", action="store_true",
                    help="N = 1 only: still create the process group (one rank) and run the N > 1 line's train leg with its "
                         "all-reduce -- the rehearsal of the RCCL calls that a one-GPU box allows")
    ap.add_argument("--physics-substeps", type=int, default=1,
                    help="physics workload: RK4 sub-steps of 1 ms fused per launch (1 = Simplified6DOF.step(0.01); 20 = the env's backend step)")
    ap.add_argument("--ppo-steps", type=int, default=16, help="rollout length per PPO iteration (train workload)")
    ap.add_argument("--ppo-epochs", type=int, default=2)
    ap.add_argument("--ppo-minibatches", type=int, default=2)
    return ap.parse_args()


def setup_dist(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # a scaling point must be what it says it is: never print "n_gpus": 1 for a `--gpus 8` request or vice versa
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher environment says WORLD_SIZE={world}; start it as "
                         f"`python bench.py --gpus {args.gpus}` (self-launching) or `python -m torch.distributed.run "
                         f"--nproc-per-node {args.gpus} bench.py --gpus {args.gpus}`")
    dev = args.device_index if args.device_index >= 0 else local
    torch.cuda.set_device(dev)
    if world > 1 or args.single_rank_group:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
            os.environ["HCRL_FORCE_COLLECTIVE"] = "1"         # ppo.FlatGrad: issue the all-reduce in a one-rank group too
        # a rank that dies leaves the others in a collective: bound the wait (the default is 10 minutes per call)
        tmo = datetime.timedelta(seconds=240)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev), timeout=tmo)
        else:
            dist.init_process_group("gloo", timeout=tmo)
    return world, rank, local


class Workload:
    """Builds the device state once; `step(k)` enqueues ONE launch on the current stream."""

    def __init__(self, args, rank):
        from hcrl_amd import config as cfgmod
        from hcrl_amd.fleet import BatchedCascade, BatchedSixDOF
        from hcrl_amd.flight_types import ControllerConfig
        from hcrl_amd.rate_env import GpuRateVecEnv
        n, prec, w = args.batch, args.precision, args.workload
        self.kind, self.n = w, n
        seed = 1000 * rank
        dev = torch.device("cuda", torch.cuda.current_device())
        if w in ("rollout", "train"):
            from hcrl_amd.policy import RateLSTMPolicy
            from hcrl_amd.ppo import PPOConfig, RecurrentPPO
            self.env = GpuRateVecEnv(n, "easy", 10.0, 0.02, "step", seed=seed, precision=prec, sampling="device")
            pol = RateLSTMPolicy(compute_dtype=torch.bfloat16 if args.policy_dtype == "bf16" else None)
            cfg = PPOConfig(n_steps=args.ppo_steps, n_epochs=args.ppo_epochs, n_minibatches=args.ppo_minibatches)
            self.ppo = RecurrentPPO(self.env, pol, cfg, seed=42)
            self.policy_flops = pol.flops_per_env_step()
            pol.prepare_inference()
            if w == "rollout":
                self.units_per_step = n
                self.desc = (f"PPO rollout step: LSTM policy inference ({args.policy_dtype} GEMMs, {pol.num_parameters()} params) "
                             f"+ fused rate_env_step, {n} envs/GPU, easy/step")
            else:
                self.units_per_step = n * args.ppo_steps
                self.desc = (f"full PPO iteration: {args.ppo_steps}-step rollout + GAE + {args.ppo_epochs} epochs x "
                             f"{args.ppo_minibatches} minibatches BPTT ({args.policy_dtype}), flat-gradient all-reduce, {n} envs/GPU")
        elif w in ("env", "env_pid"):
            self.env = GpuRateVecEnv(n, "medium", 10.0, 0.02, "step", seed=seed, precision=prec, sampling="device")
            self.env.reset()
            g = torch.Generator(device=dev).manual_seed(seed + 1)
            # 8 different resident action batches, cycled: smooth-ish bounded surfaces around trim
            self.actions = [torch.cat([(torch.rand((n, 3), device=dev, generator=g) - 0.5) * 0.6,
                                       0.4 + 0.4 * torch.rand((n, 1), device=dev, generator=g)], 1).contiguous()
                            for _ in range(8)]
            self.units_per_step = n
            self.desc = f"rate_env_step fused (20 RK4 sub-steps + reward/obs/done/auto-reset), {n} envs/GPU, medium/step"
        elif w == "physics":
            # SURVEY §8d cfg 2: FlightEnvelopeSampler-distributed ICs, fixed controls, one RK4 of 10 ms per launch
            rs = np.random.RandomState(seed)
            x0 = np.zeros((n, 12))
            x0[:, 3] = rs.uniform(15, 30, n); x0[:, 2] = -rs.uniform(50, 200, n)
            x0[:, 6] = rs.uniform(-0.26, 0.26, n); x0[:, 7] = rs.uniform(-0.26, 0.26, n); x0[:, 8] = rs.uniform(0, 6.28, n)
            x0[:, 9:12] = rs.uniform(-0.1, 0.1, (n, 3))
            u = np.concatenate([rs.uniform(-0.3, 0.3, (n, 3)), rs.uniform(0.3, 0.9, (n, 1))], 1)
            self.fleet = BatchedSixDOF(n, prec)
            self.fleet.reset(x0); self.fleet.set_controls(u)
            self.n_sub = int(getattr(args, "physics_substeps", 1))
            self.units_per_step = n * self.n_sub
            self.desc = (f"Simplified6DOF.step(0.01), one RK4 per launch, {n} aircraft/GPU" if self.n_sub == 1 else
                         f"SimulationAircraftBackend.step(0.02) = {self.n_sub} RK4 sub-steps of 1 ms per launch, {n} aircraft/GPU")
        else:
            fc = cfgmod.load_controller_config("cascaded_pid.yaml")
            mc = cfgmod.load_mission_config("square_pattern.yaml")
            rs = np.random.RandomState(seed)
            x0 = np.zeros((n, 12)); x0[:, 2] = -mc.altitude; x0[:, 3] = mc.speed
            x0[:, 0:2] = rs.uniform(-20, 20, (n, 2)); x0[:, 8] = rs.uniform(-0.1745, 0.1745, n)
            self.fleet = BatchedCascade(n, cfgmod.square_mission(mc.pattern_size, mc.altitude, mc.speed), prec,
                                        ControllerConfig(), fc, guidance_type=mc.guidance, on_complete="restart")
            self.fleet.reset(x0)
            self.inner = 10
            self.units_per_step = n * self.inner
            self.desc = (f"5-level cascade + 1 RK4 per 10 ms control step, {self.inner} control steps per launch, "
                         f"{n} aircraft/GPU, square mission (restarts)")

    @torch.no_grad()
    def _rollout_step(self):
        """One policy + env step with every buffer updated in place (capture-safe)."""
        p = self.ppo
        from hcrl_amd.fused import episode_flags
        nxt = p._state_bufs[1 - p._cur]
        fused_glue = p.policy._fused_ok(p.obs)
        # the previous step's done flags -> episode_start / keep / noise counter inside the step's first kernel (as RecurrentPPO._rollout_body)
        a, _v, _lp, new_states = p.policy.step(p.obs, p.states, p.episode_start, out_states=nxt, keep=p.keep, bump_noise=not fused_glue,
                                               done_flags=(self.env.terminated, self.env.truncated) if fused_glue else None)
        _obs, _r, term, trunc = self.env.step_device(a)          # p.obs aliases the env's observation buffer
        for dst, src in zip(nxt, new_states):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        p._cur = 1 - p._cur
        if not fused_glue:
            episode_flags(term, trunc, p.episode_start, p.keep, None)

    def step(self, k):
        if self.kind == "rollout":
            self._rollout_step()
        elif self.kind == "train":
            self.ppo.collect_rollout()
            self.ppo.update()
        elif self.kind == "env":
            self.env.step_device(self.actions[k % len(self.actions)])
        elif self.kind == "env_pid":
            self.env.step_device(None)
        elif self.kind == "physics":
            if self.n_sub == 1:
                self.fleet.step(0.01)
            else:
                self.fleet.step(0.001 * self.n_sub, 0.001)
        else:
            self.fleet.run(0.01, self.inner)


class StepLoop:
    """Runs `n` steps of a workload: whole hipGraph replays of `gsteps` steps first, then eager leftovers.

    The env's event counters (`GpuRateVecEnv._ev_slot`) and the policy's recurrent-state buffers (`RecurrentPPO._cur`)
    ping-pong once per step on the HOST, and a captured graph bakes in the buffer pointers of the parity it was captured
    at.  gsteps is even, so a replay preserves parity, but an odd eager tail flips it: one graph is therefore captured
    per parity (before any timing) and the loop replays the one that matches the number of steps executed so far."""

    def __init__(self, wl, args):
        self.wl, self.count, self.graphs, self.gsteps = wl, 0, {}, 0
        K = args.steps
        if args.graph and K >= 2 and args.workload in ("env", "env_pid", "physics", "cascade", "rollout"):
            self.gsteps = max(2, min(args.graph_steps, K) // 2 * 2)
            stream = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(stream)
            with torch.cuda.stream(side):
                for _ in range(4):
                    self._eager()                                  # warm the allocator / lazy init before capture
            stream.wait_stream(side)
            torch.cuda.synchronize()
            for _ in range(2):
                g = torch.cuda.CUDAGraph()
                # thread-local capture mode: with N > 1 the process group's watchdog thread may touch the HIP runtime while
                # this thread captures; only calls made by the capturing thread should be able to invalidate the capture
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    for _ in range(self.gsteps):
                        self._eager()
                self.count -= self.gsteps                          # captured, not executed
                self.graphs[self.count % 2] = g
                self._eager()                                      # one real step: the other parity
            torch.cuda.synchronize()

    def _eager(self):
        self.wl.step(self.count)
        self.count += 1

    def run(self, nsteps):
        done = 0
        if self.graphs:
            g = self.graphs[self.count % 2]
            while nsteps - done >= self.gsteps:
                g.replay()
                done += self.gsteps
                self.count += self.gsteps
        for _ in range(nsteps - done):
            self._eager()

    @property
    def mode(self):
        return ("hipGraph x%d" % self.gsteps) if self.graphs else "eager"


def _grouped():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def timed_region(wl, args, world, min_region_s=0.05, max_repeats=31):
    """W warm-up steps, then EXACTLY K timed steps bracketed by barrier + synchronize; HIP events on the launch stream.
    When the K-step region is shorter than `min_region_s` (K = 20 launches of 68 us is 1.4 ms: two timer reads and a
    launch-queue hiccup are a tenth of it) the same K-step region is timed R times and the MEDIAN is reported
    (`repeats` in the output); K itself never changes."""
    K, W = args.steps, args.warmup
    grp = world > 1 or (getattr(args, "single_rank_group", False) and _grouped())
    stream = torch.cuda.current_stream()
    loop = StepLoop(wl, args)
    loop.run(W)

    def once():
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if grp:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record(stream)
        loop.run(K)
        ev1.record(stream)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0                # this rank's K steps; the MAX over ranks below is the job's time
        if grp:
            torch.distributed.barrier()                # closing bracket: nobody leaves (or starts the next repeat) early --
        dev_ms = ev0.elapsed_time(ev1)                 # its own latency (an all-reduce + sync) is not part of the K steps
        if grp:
            t = torch.tensor([wall], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            wall = float(t.item())
        return wall, dev_ms

    samples = [once()]
    repeats = 1
    if samples[0][0] < min_region_s:
        repeats = int(min(max_repeats, max(3, (0.25 / max(samples[0][0], 1e-6)))) // 2 * 2 + 1)      # odd
        if grp:                                              # every rank must take the same number of barriers
            t = torch.tensor([repeats], dtype=torch.int64, device="cuda" if args.backend == "nccl" else "cpu")
            torch.distributed.broadcast(t, 0)
            repeats = int(t.item())
        samples += [once() for _ in range(repeats - 1)]
    walls = sorted(s[0] for s in samples)
    devs = sorted(s[1] for s in samples)
    return walls[len(walls) // 2], devs[len(devs) // 2], loop.mode, {"repeats": repeats, "wall_min_s": walls[0], "wall_max_s": walls[-1]}


def train_leg(args, rank, world):
    """The RCCL-exercising leg of the N > 1 line (and the `train` extra at N = 1): a few full PPO iterations -- rollout
    graph, GAE, epochs x env-slices of BPTT, ONE flat all-reduce of the 7.3 MB gradient buffer per optimizer step -- with
    the collective timed by HIP events on the stream that waits for it."""
    import copy
    a = copy.copy(args)
    a.workload = "train"
    grp = world > 1 or (getattr(args, "single_rank_group", False) and _grouped())
    wl = Workload(a, rank)
    flat = wl.ppo.flat
    iters, warm = 3, 1
    for k in range(warm):
        wl.step(k)
    torch.cuda.synchronize()
    flat.events = []
    if grp:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(iters):
        wl.step(k)
    torch.cuda.synchronize()
    if grp:
        torch.distributed.barrier()
    wall = time.perf_counter() - t0
    ar_ms = [e0.elapsed_time(e1) for e0, e1 in flat.events]
    flat.events = None
    if grp:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        wall = float(t.item())
    opt_steps = a.ppo_epochs * a.ppo_minibatches
    res = {"value": wl.units_per_step * iters * world / wall, "unit": "env-steps/s", "ms_per_iteration": wall * 1e3 / iters,
           "iterations": iters, "workload": wl.desc, "optimizer_steps_per_iteration": opt_steps,
           "policy": policy_block(wl, a, wall / iters, "train")}
    if grp:
        res["collective"] = {"op": "all_reduce(sum) of the flat fp32 gradient buffer", "backend": args.backend,
                             "library": "RCCL over xGMI" if args.backend == "nccl" else "gloo (CPU rehearsal)",
                             "n_ranks_in_group": torch.distributed.get_world_size(), "bytes": int(flat.buf.numel() * 4),
                             "calls_timed": len(ar_ms), "allreduce_us_per_optimizer_step": 1e3 * sum(ar_ms) / max(len(ar_ms), 1),
                             "allreduce_us_min": 1e3 * min(ar_ms) if ar_ms else None,
                             "allreduce_us_max": 1e3 * max(ar_ms) if ar_ms else None,
                             "note": "HIP events on the compute stream around the call: includes waiting for the slowest rank"}
    del wl
    torch.cuda.empty_cache()
    return res


def policy_block(wl, args, s_per_step, kind):
    """MFMA side of a rollout / train workload: achieved policy TFLOP/s against the dense bf16 MFMA peak, plus the
    MFMA-pipe occupancy of the recurrent cell from the committed PMC run (profiles/traffic.json)."""
    mult = 1.0 if kind == "rollout" else (1.0 + 3.0 * args.ppo_epochs)      # fwd, or fwd + epochs x (fwd + bwd)
    pf = wl.policy_flops * mult * wl.units_per_step / s_per_step / 1e12
    peak = 2500.0 if args.policy_dtype == "bf16" else 157.3
    blk = {"flops_per_env_step_fwd": wl.policy_flops, "achieved_tflops": pf, "dtype": args.policy_dtype,
           "peak_tflops_dense": peak, "frac": pf / peak}
    tpath = os.path.join(REPO, "profiles", "traffic.json")
    if os.path.exists(tpath):
        blk["lstm_cell_pmc"] = json.load(open(tpath)).get("lstm_mfma_65536")
    return blk


def measured_drift(precision):
    """Percentiles of the per-aircraft / per-env error against the CPU oracle, as measured by tests/test_gpu_parity_scale.py on
    the GPU box (4096 distinct aircraft x 1000 steps, 4096 envs x 520 steps) and committed as profiles/drift.json."""
    dpath = os.path.join(REPO, "profiles", "drift.json")
    if not os.path.exists(dpath):
        return None
    d = json.load(open(dpath)).get(precision)
    if not d:
        return None
    keep = ("p50", "p90", "p99", "p99.9", "max", "n", "n_regular", "max_regular", "n_over_1e-4", "n_over_gate",
            "fp32_argument_model_p50", "fp32_argument_model_p99", "n_model_over_1e-4", "flag_mismatch_envs", "max_where_d7_le_1e5",
            "n_reproducible", "max_where_reproducible", "n_amp_gt_1e3", "n_oracle_twins_jump", "n_over_its_bound")
    return {k: {q: v[q] for q in keep if q in v} for k, v in d.items() if isinstance(v, dict)}


def extras(args):
    """Short secondary measurements (same GPU, same batch) reported beside the headline; each ~1-3 s."""
    import copy
    res = {}
    # "*_saturation": the same kernels at a batch that fills the chip many times over ("report physics-only
    # saturation throughput"): 4 Mi aircraft / 1 Mi envs per GPU -- what the hardware sustains once launch cost is amortised
    for key, wl_name, steps, warm, batch, prec in (
            ("env_f64", "env", 100, 10, None, "f64"), ("env_f32", "env", 200, 20, None, "f32"),
            ("physics", "physics", 400, 40, None, None), ("physics_f64", "physics", 200, 20, None, "f64"),
            ("physics_f32", "physics", 400, 40, None, "f32"),
            ("cascade", "cascade", 100, 10, None, None),
            ("rollout", "rollout", 100, 10, None, None),
            ("physics_20_substeps", "physics", 200, 20, None, None),
            ("physics_saturation", "physics", 100, 10, 1 << 22, None),
            ("env_saturation", "env", 60, 6, 1 << 20, None)):
        try:
            a = copy.copy(args)
            a.workload, a.steps, a.warmup = wl_name, steps, warm
            if batch is not None:
                a.batch = batch
            if prec is not None:
                a.precision = prec
            a.physics_substeps = 20 if key == "physics_20_substeps" else 1
            wl = Workload(a, 0)
            # the rollout leg is the one whose 100-step region (26 ms) one launch-queue hiccup moves by several per cent: median of repeats
            wall, dev_ms, mode, _rep = timed_region(wl, a, 1, min_region_s=0.2 if key == "rollout" else 0.0, max_repeats=9)
            res[key] = {"value": wl.units_per_step * steps / wall, "ms_per_step": wall * 1e3 / steps,
                        "unit": "env-steps/s" if wl_name in ("rollout", "env") else "aircraft-steps/s",
                        "workload": wl.desc}
            if prec is not None:
                res[key]["precision"] = prec
            if key in ("env_f64", "env_f32"):
                res[key]["drift_vs_oracle"] = measured_drift(prec)
            if key in ("physics", "physics_f64", "physics_f32", "cascade"):
                # these legs step at dt = 10 ms (Simplified6DOF.step(0.01), examples/03's control step): the drift that belongs
                # to them is the 10 ms row -- for `mixed` it holds the 1e-4 gate only on the aircraft clear of the reference's
                # guards (n_regular / max_regular), not per aircraft as the 1 ms row does; `physics_f64` is the variant that does
                d = (measured_drift(prec or a.precision) or {}).get("cfg2_dt0.01_1000_steps")
                res[key]["precision"] = prec or a.precision
                res[key]["dt_s"] = 0.01
                res[key]["drift_vs_oracle_dt10ms"] = d
            if key == "rollout":
                res[key]["policy"] = policy_block(wl, a, wall / steps, "rollout")
                res[key]["repeats"] = _rep
            del wl
            torch.cuda.empty_cache()
        except Exception as ex:
            res[key] = {"error": repr(ex)}
    for key, epochs in (("train", args.ppo_epochs), ("train_10_epochs", 10)):
        # the second one is the reference's epoch count (learned_controllers/config/ppo_lstm.yaml:38-58: n_epochs 10)
        try:
            a = copy.copy(args)
            a.ppo_epochs = epochs
            res[key] = train_leg(a, 0, 1)
        except Exception as ex:
            res[key] = {"error": repr(ex)}
    return res


def usable_cores(omp_max):
    """Host cores this process may actually run on: OpenMP's count capped by the affinity mask and the cgroup CPU quota
    (a GPU box hands a 1-GPU job a share of the host; one thread per *visible* core oversubscribes that share)."""
    n = int(omp_max)
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            parts = open(path).read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args):
    """The CPU oracle (test infrastructure) timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import oracle as orc
    from hcrl_amd import layout as L, samplers
    from hcrl_amd.params import AircraftParams
    threads = usable_cores(orc.lib.orc_max_threads())
    n = 4096
    P, EC = AircraftParams().to_block(), samplers.env_consts("medium", 10.0, 0.02, "step")
    rs = np.random.RandomState(0)
    x = np.zeros((12, n)); x[3] = rs.uniform(15, 30, n); x[2] = -rs.uniform(50, 200, n)
    x[6] = rs.uniform(-0.26, 0.26, n); x[7] = rs.uniform(-0.26, 0.26, n); x[8] = rs.uniform(0, 6.28, n)
    e = np.zeros((L.FD_NE, n)); e[L.FD_E_PREV_THR] = 0.5; e[L.FD_E_CMD_P] = rs.uniform(-0.8, 0.8, n)
    ei = np.zeros((L.FD_NEI, n), np.int32)
    acts = np.concatenate([(rs.rand(n, 3) - 0.5) * 0.6, 0.4 + 0.4 * rs.rand(n, 1)], 1).astype(np.float32)
    obs = np.zeros((n, 18), np.float32); rew = np.zeros(n); te = np.zeros(n, np.int32); tr = np.zeros(n, np.int32)

    def one(nthreads):
        if args.workload == "physics":
            u = np.ascontiguousarray(np.concatenate([acts[:, 1:2], acts[:, 0:1], acts[:, 2:4]], 1).T.astype(np.float64))
            orc.lib.orc_sixdof_step_batch(orc.dp(P), orc.dp(x), orc.dp(u), n, 0.01, 1, nthreads)
        else:
            orc.lib.orc_env_step_batch(orc.dp(P), orc.dp(EC), orc.dp(x), orc.dp(e), orc.ip(ei), orc.fp(acts), orc.fp(obs),
                                       orc.dp(rew), orc.ip(te), orc.ip(tr), n, nthreads)
    # pick the thread count this box actually sustains (its CPU share is usually far below the visible core count):
    # ~0.5 s per candidate, best rate wins
    omp_max = int(orc.lib.orc_max_threads())
    best = (0.0, threads)
    for cand in sorted({threads, *[c for c in (8, 16, 32, 64) if c <= omp_max], omp_max}):
        one(cand)
        t0, r = time.perf_counter(), 0
        while r < 2 or time.perf_counter() - t0 < 0.5:
            one(cand); r += 1
        rate = r / (time.perf_counter() - t0)
        ei[L.FD_EI_STEP] = 0
        if rate > best[0]:
            best = (rate, cand)
    threads = best[1]
    t0, reps = time.perf_counter(), 0
    while time.perf_counter() - t0 < args.cpu_seconds:
        one(threads)
        reps += 1
        ei[L.FD_EI_STEP] = 0                      # keep every env alive and un-truncated: constant work per step
    dt_all = time.perf_counter() - t0
    t0 = time.perf_counter()
    one(1)
    dt_one = time.perf_counter() - t0
    unit = "aircraft-steps/s" if args.workload == "physics" else "env-steps/s"
    return {"value": n * reps / dt_all, "unit": unit, "cores": threads, "kind": "port",
            "single_core_value": n / dt_one,
            # context, not measured in this run: the reference's own Python env step, timed by the survey in the build
            # container (RateControlEnv.step, 214 env-steps/s on one core; the reference cannot travel)
            "reference_python_single_core": 214.0 if unit == "env-steps/s" else None,
            "cpu_model": cpu_model(), "visible_cores": omp_max,
            "sample": f"oracle/flight_oracle.c (fp64, OpenMP), {n} envs x {reps} steps, {dt_all:.1f} s on {threads} threads"}


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON: whatever libraries print there (gloo's "[Gloo] Rank ..." lines, RCCL
    # warnings) is sent to stderr for the whole run, and the result is written to the saved descriptor at the end
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    world, rank, _ = setup_dist(args)
    wl = Workload(args, rank)
    wall, dev_ms, mode, rep = timed_region(wl, args, world)
    K = args.steps
    units = wl.units_per_step * K * world
    value = units / wall
    desc, units_per_step = wl.desc, wl.units_per_step
    pol = policy_block(wl, args, wall / K, args.workload) if args.workload in ("rollout", "train") else None
    del wl
    torch.cuda.empty_cache()
    train = None
    if (world > 1 or args.single_rank_group) and args.workload == "env" and not args.no_extras:
        # every rank runs it: it contains the collective.  A failure must not cost the headline line (a deterministic error
        # reaches every rank alike; a rank lost inside a collective ends the others at the process group's timeout)
        try:
            train = train_leg(args, rank, world)
        except Exception as ex:
            train = {"error": repr(ex)}
    if rank != 0:
        return
    per_launch_s = dev_ms * 1e-3 / K
    alg_bytes = ALG_BYTES[args.workload] * units_per_step
    achieved = alg_bytes / per_launch_s / 1e9
    traffic, valu = None, None
    tpath = os.path.join(REPO, "profiles", "traffic.json")           # PMC-derived HBM bytes per launch, if collected
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        traffic = tj.get(f"{args.workload}_{args.precision}_{args.batch}")
        valu = tj.get(f"{args.workload}_{args.precision}_{args.batch}_valu")      # VALU-pipe occupancy from the same PMC runs
    tflops = ALG_FLOPS[args.workload] * units_per_step / per_launch_s / 1e12
    valu_bound = args.workload in ("env", "env_pid", "rollout", "train") or (args.workload == "physics" and args.physics_substeps > 1)
    out = {
        "metric": "env-steps/sec (whole node), rate-control task, batch 65536 per GPU" if args.workload.startswith("env")
                  else f"{args.workload} {'env' if args.workload in ('rollout', 'train') else 'aircraft'}-steps/sec",
        "value": value, "unit": "env-steps/s" if args.workload in ("env", "env_pid", "rollout", "train") else "aircraft-steps/s",
        "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": wall * 1e3 / K,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"f64": "f64", "mixed": "f32 compute / f64 state", "f32": "f32"}[args.precision],
        "data": "synthetic",
        "parity": {"variant": args.precision, "drift_vs_oracle": measured_drift(args.precision),
                   "source": "tests/test_gpu_parity_scale.py on MI355X -> profiles/drift.json"},
        "repeats": rep["repeats"], "region_wall_s": {"median": wall, "min": rep["wall_min_s"], "max": rep["wall_max_s"]},
        "config": {"workload": desc, "precision": args.precision, "batch_per_gpu": args.batch,
                   "launch": mode, "parallelism": f"{world} independent env shards, no data-path collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
                     "kernel_ms": per_launch_s * 1e3,
                     "binding": "valu" if valu_bound else "hbm",
                     "valu": {"achieved": tflops, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": tflops / FP32_VECTOR_PEAK_TFLOPS,
                              "flops_per_unit": ALG_FLOPS[args.workload]},
                     "note": ("the fused env step is vector-ALU bound (AI ~ 133 flop/B): `frac` is the HBM fraction the "
                              "north-star asks for, `valu.frac` is the binding roof") if valu_bound else
                             "HBM-bound launch; valu.frac reported beside it"},
        "compute": {"achieved_tflops": tflops, "peak_tflops": FP32_VECTOR_PEAK_TFLOPS,
                    "frac": tflops / FP32_VECTOR_PEAK_TFLOPS, "valu_pmc": valu},
    }
    if pol is not None:
        out["policy"] = pol
    if train is not None:
        out["train"] = train
    if world == 1 and args.workload == "env" and not args.no_extras and not args.single_rank_group:
        out["extras"] = extras(args)
    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(args)
        except Exception as ex:  # the oracle is optional test infrastructure; never fail the measurement over it
            out["cpu_baseline"] = {"value": None, "error": repr(ex)}
    sys.stdout.flush()
    os.write(result_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
