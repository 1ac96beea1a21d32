"""Benchmark of the MI355X hot path.

  python bench.py --gpus N --steps K --warmup W [--workload c2|c4|c5shard|c5]

A step = one pass of the fused residual + Jacobian + Schur linearisation (vs_ba_linearize) over
one synthetic bundle-adjustment problem resident in HBM, followed for N > 1 by the exchange that
sums the ranks' partial reduced camera systems (vs_ba_exchange_reduced).  The default workload is BASELINE.json configs[3] (1k cameras, 500k
landmarks, ~2M observations), the shape the metric's "1-8 GPU" part is quoted on: it fits one GPU
and its landmarks shard over N GPUs, so every N runs the SAME problem (strong scaling): rank r
linearises landmark shard r and sends the rows of the reduced system its landmarks touch; one
RCCL all-gather issued by the library on the step's own stream moves them and every rank adds the
pieces in rank order (the all-reduce of the whole buffer this replaced is timed beside it).
`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts N ranks itself (one process per
GPU, before anything touches a GPU here); under torch.distributed.run the ranks are taken from the
environment.  Rank 0 prints ONE JSON line; `value` is observations/s over all ranks.  The line
carries `roofline` (HIP-event kernel times measured here), `cpu_baseline` (the CPU oracle timed on
this host) and, in `extra`, the matcher leg (configs[2]), configs[1] and the larger BA shapes.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA spec


class _DevArray:
    """Exposes a raw device pointer to torch through __cuda_array_interface__."""

    def __init__(self, ptr, n, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False),
                                         "version": 2}


def trace(msg):
    """Progress lines on stderr when VS_BENCH_TRACE is set (stdout carries only the JSON line)."""
    if os.environ.get("VS_BENCH_TRACE"):
        sys.stderr.write("[bench rank %s +%.1fs] %s\n" % (os.environ.get("RANK", "0"), time.time() - _T0, msg))
        sys.stderr.flush()


_T0 = time.time()


def fused_bytes(No, Np, Nc, nblocks):
    """Algorithmic bytes of one fused linearisation in the fp64 layout the kernels use:
    in : per observation camera index 4 + point index 4 + (u,v) 16; 24 per point; 48 per camera
    out: per landmark inverse of V (sym 6) 48 + g_p 24; 288 per stored 6x6 block of S; rhs 48/camera."""
    return 24 * No + 24 * Np + 48 * Nc + 72 * Np + 288 * nblocks + 48 * Nc


def make_problem(name, rank, world, seed=0):
    from visual_slam_amd import synth
    from visual_slam_amd.ba import BAProblem
    Nc, Np, k, band = synth.BA_CONFIGS[name]
    sc = synth.make_ba_scene(Nc, Np, k, band=band, seed=seed)
    ba = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv, rank=rank,
                               world=world)
    return sc, ba


def bench_ba(args, rank, world, dist, torch, comm):
    t0 = time.time()
    sc, ba = make_problem(args.workload, rank, world)
    stream = torch.cuda.current_stream()
    ba.set_stream(stream.cuda_stream)
    # LM damping vectors resident in HBM (device pointers are used in place by the library)
    dc = torch.full((6 * sc.Nc,), 1e-4, dtype=torch.float64, device="cuda")
    dp = torch.full((3 * sc.Np,), 1e-4, dtype=torch.float64, device="cuda")
    import ctypes as C
    lib = ba.lib
    pc = C.cast(C.c_void_p(dc.data_ptr()), C.POINTER(C.c_double))
    pp = C.cast(C.c_void_p(dp.data_ptr()), C.POINTER(C.c_double))
    assert lib.vs_ba_linearize(ba.h, pc, pp) == 0
    ptr, n = ba.reduce_buffer()
    red = torch.as_tensor(_DevArray(ptr, n), device="cuda") if (dist is not None and comm is None) else None
    setup_s = time.time() - t0
    sptr = C.c_void_p(stream.cuda_stream)
    trace("problem ready (%.1f s)" % setup_s)

    # N > 1: the ranks combine their partial reduce buffers with vs_ba_exchange_reduced: every rank
    # sends only the rows its landmarks touch (one all-gather, summed in rank order).  With the
    # library's communicator the all-gather is RCCL on the step's stream; otherwise the gather and
    # sum callbacks go through torch.distributed (rehearsal backends included).
    if dist is not None:
        if comm is not None:
            assert lib.vs_ba_set_comm(ba.h, comm) == 0
        else:
            from visual_slam_amd import dist as vdist
            if dist.get_backend() == "nccl":
                vdist.attach_allgather(ba)
                vdist.attach_allreduce(ba)
            else:                   # rehearsal backend (gloo): through the host
                def gather_host(send_ptr, recv_ptr, m):
                    send = vdist.device_tensor(send_ptr, m).cpu()
                    parts = [torch.empty_like(send) for _ in range(world)]
                    dist.all_gather(parts, send)
                    vdist.device_tensor(recv_ptr, m * world).copy_(torch.cat(parts))
                    torch.cuda.synchronize()

                def sum_host(buf_ptr, m):
                    t = vdist.device_tensor(buf_ptr, m)
                    host = t.cpu()
                    dist.all_reduce(host)
                    t.copy_(host)
                    torch.cuda.synchronize()
                ba.set_allgather(gather_host)
                ba.set_allreduce(sum_host)

    def exchange():
        if dist is not None:
            assert lib.vs_ba_exchange_reduced(ba.h) == 0

    def exchange_whole_buffer():    # the all-reduce of the whole buffer this replaced, for comparison
        os.environ["VS_BA_EXCHANGE"] = "allreduce"
        try:
            assert lib.vs_ba_exchange_reduced(ba.h) == 0
        finally:
            del os.environ["VS_BA_EXCHANGE"]

    def step():
        rc = lib.vs_ba_linearize(ba.h, pc, pp)
        assert rc == 0
        exchange()

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, reps):
        sync()
        t1 = time.perf_counter()
        for _ in range(reps):
            fn()
        sync()
        dt = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([dt], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    for _ in range(args.warmup):
        step()
    trace("warm-up done")
    dt = timed(step, args.steps)
    trace("timed region done")
    allreduce_ms = exchange_ms = None
    send_doubles = recv_doubles = 0
    if dist is not None:
        reps = max(10, args.steps // 10)
        exchange_ms = timed(exchange, reps) / reps * 1e3
        allreduce_ms = timed(exchange_whole_buffer, reps) / reps * 1e3
        send_doubles, recv_doubles = ba.exchange_size()
    ms_per_step = dt / args.steps * 1e3
    total_obs = sc.No
    value = total_obs / (dt / args.steps)
    out = dict(ms_per_step=ms_per_step, value=value, setup_s=setup_s, No=int(sc.No), Np=int(sc.Np),
               Nc=int(sc.Nc), allreduce_ms_per_step=allreduce_ms, allreduce_bytes=int(n) * 8,
               exchange_ms_per_step=exchange_ms, exchange_send_bytes=int(send_doubles) * 8,
               exchange_recv_bytes=int(recv_doubles) * 8)
    if rank == 0:
        ms = ba.profile_linearize(dc.cpu().numpy(), dp.cpu().numpy(), iters=max(20, args.steps // 2))
        br, _ = ba.block_index()
        # this rank's shard: its observations and landmarks, every camera and block
        alg = fused_bytes(sc.No // world, sc.Np // world, sc.Nc, len(br))
        k_ms = kernel_times(ms)
        kern_sum = float(sum(k_ms.values()))
        dom = max(k_ms, key=k_ms.get)
        achieved = alg / (kern_sum * 1e-3) / 1e9
        traffic, traffic_src = committed_traffic(args.workload) if world == 1 else (None, None)
        # the block kernel is bound by fp64 issue, not by HBM: its VALU wave-instructions (committed
        # counter pass) at one per 4 cycles and SIMD, against its measured duration
        valu = committed_valu_instructions(args.workload) if world == 1 else None
        props = torch.cuda.get_device_properties(torch.cuda.current_device())
        clock_hz = float(getattr(props, "clock_rate", 0) or 2.4e6) * 1e3
        valu_frac = None
        if valu and k_ms.get("blocks(camera+camera-pair)"):
            ideal_s = valu * 4.0 / (props.multi_processor_count * 4 * clock_hz)
            valu_frac = ideal_s / (k_ms["blocks(camera+camera-pair)"] * 1e-3)
        out["roofline"] = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                               frac=achieved / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_src,
                               fp64_valu_frac=valu_frac,
                               fp64_valu_note="block kernel: SQ_INSTS_VALU (committed counter pass) x 4 cycles / "
                                              "(SIMDs x %.2f GHz) / its measured duration" % (clock_hz / 1e9),
                               kernel="fused linearise pipeline: camera records, landmark kernel, block kernels, "
                                      "and finalize when a block spans several work items. HIP-event intervals, "
                                      "each carrying ~5 us of event overhead, so the sum is longer than the timed step",
                               dominant_kernel=dom, kernels_ms=k_ms, pipeline_ms=float(ms[5]),
                               algorithmic_bytes=int(alg),
                               # the same bytes over the dominant kernel's own HIP-event duration
                               frac_dominant_kernel=(alg / (k_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS) if k_ms.get(dom) else None)
    if rank == 0 and world == 1 and not args.no_extra:
        # throughput with two independent problems in flight (a second handle on a second stream,
        # steps alternating): one problem's landmark kernel runs beside the other's block kernel.
        # Reported beside the headline, which stays one problem on one stream (an LM loop is serial).
        s2 = torch.cuda.Stream()
        sc2, ba2 = make_problem(args.workload, rank, world)
        ba2.set_stream(s2.cuda_stream)
        pair = ((ba, pc, pp), (ba2, pc, pp))
        for it in range(2 * args.warmup):
            b_, c_, p_ = pair[it % 2]
            assert lib.vs_ba_linearize(b_.h, c_, p_) == 0
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for it in range(2 * args.steps):
            b_, c_, p_ = pair[it % 2]
            lib.vs_ba_linearize(b_.h, c_, p_)
        torch.cuda.synchronize()
        per2 = (time.perf_counter() - t2) / (2 * args.steps)
        out["two_streams"] = dict(ms_per_step=per2 * 1e3, obs_per_s=sc.No / per2,
                                  note="two problems of this workload on two streams, steps alternating")
        ba2.close()
    ba.close()
    return out


def committed_traffic(workload):
    """HBM bytes per linearise pass from the committed rocprofv3 PMC passes under profiles/
    (FETCH_SIZE and WRITE_SIZE are kilobytes; FETCH_SIZE is doubled as the gfx950 counter tallies
    128-byte requests of 16-byte-per-lane reads at 64 bytes).  None when no summary is present."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r[0-9][0-9]_summary.json")))
    if not files:
        return None, None
    pmc = json.load(open(files[-1])).get("pmc_avg_per_dispatch", {})
    f, w = pmc.get(workload + "_fetch"), pmc.get(workload + "_write")
    if not f or not w:
        return None, None
    keys = ("prep_cameras_kernel", "landmark_", "blocks_", "finalize_kernel")
    fetch = sum(v["FETCH_SIZE"] for k, v in f.items() if any(x in k for x in keys))
    write = sum(v["WRITE_SIZE"] for k, v in w.items() if any(x in k for x in keys))
    return int((2.0 * fetch + write) * 1024), os.path.basename(files[-1])


def committed_valu_instructions(workload):
    """VALU wave-instructions of the block kernels per linearise pass (SQ_INSTS_VALU of the committed
    rocprofv3 PMC pass), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r[0-9][0-9]_summary.json")))
    if not files:
        return None
    v = json.load(open(files[-1])).get("pmc_avg_per_dispatch", {}).get(workload + "_valu")
    if not v:
        return None
    n = sum(c.get("SQ_INSTS_VALU", 0.0) for k, c in v.items() if "blocks_" in k)
    return n or None


def kernel_times(ms):
    """vs_ba_profile_linearize's stage averages as a dict (slot 3 is not used by the library)."""
    names = ["prep_cameras", "landmark", "blocks(camera+camera-pair)", None, "finalize"]
    return {nm: float(v) for nm, v in zip(names, ms[:5]) if nm}


def bench_ba_shape(name, torch, steps=20):
    """Single-GPU roofline of a larger BA shape (reported under extra)."""
    sc, ba = make_problem(name, 0, 1)
    ba.set_stream(torch.cuda.current_stream().cuda_stream)
    dc = np.full(6 * sc.Nc, 1e-4); dp = np.full(3 * sc.Np, 1e-4)
    ms = ba.profile_linearize(dc, dp, iters=steps)
    br, _ = ba.block_index()
    alg = fused_bytes(sc.No, sc.Np, sc.Nc, len(br))
    k_ms = kernel_times(ms)
    kern = float(sum(k_ms.values()))
    # wall time of back-to-back steps with the damping vectors resident in HBM, as the headline
    import ctypes as C
    dcd = torch.full((6 * sc.Nc,), 1e-4, dtype=torch.float64, device="cuda")
    dpd = torch.full((3 * sc.Np,), 1e-4, dtype=torch.float64, device="cuda")
    pc = C.cast(C.c_void_p(dcd.data_ptr()), C.POINTER(C.c_double))
    pp = C.cast(C.c_void_p(dpd.data_ptr()), C.POINTER(C.c_double))
    n_wall = max(steps, min(500, int(0.05 / max(kern * 1e-3, 1e-6))))
    for _ in range(5):
        assert ba.lib.vs_ba_linearize(ba.h, pc, pp) == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_wall):
        ba.lib.vs_ba_linearize(ba.h, pc, pp)
    torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t0) / n_wall * 1e3
    ba.lm_step(1e4, fetch=False)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(3):
        ba.lm_step(1e4, fetch=False)     # the step stays on the device, as inside vs_ba_solve
    torch.cuda.synchronize()
    lm_ms = (time.perf_counter() - t1) / 3 * 1e3
    ba.close()
    return dict(workload=name, No=int(sc.No), Np=int(sc.Np), Nc=int(sc.Nc), n_blocks=int(len(br)),
                obs_per_s=sc.No / (step_ms * 1e-3), ms_per_step=step_ms, steps=n_wall, kernels_ms=k_ms,
                lm_iteration_ms=lm_ms,
                roofline=dict(bound="hbm", achieved=alg / (kern * 1e-3) / 1e9, peak=HBM_PEAK_GBS,
                              unit="GB/s", frac=alg / (kern * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              algorithmic_bytes=int(alg)))


def bench_solve_small():
    """Whole vs_ba_solve runs at the reference's own operating point (config 1: 10 keyframes x 500
    points, the call order of Tracking::solveCamera / solvePoints, src/core/tracking.cc:79-163), and
    at config 2 (the CPU oracle's LM loop on the same problems: cpu_baseline.solve)."""
    from visual_slam_amd import synth
    from visual_slam_amd.ba import BAProblem
    out = {}

    def setup_ms(sc, pts, fix):
        """Median time of building a problem from host arrays (vs_ba_create, set_cameras, set_points,
        set_observations) and destroying it: the reference builds a fresh Optimizer per solve."""
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            g = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, pts, sc.cam, sc.pt, sc.uv, fix_points=fix)
            g.close()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts[2:])) * 1e3
    for name in ("c1", "c2"):
        sc = synth.make_ba_config(name, seed=0)
        for fix in (True, False):
            pts = sc.pts_gt if fix else sc.pts0
            gpu = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, pts, sc.cam, sc.pt, sc.uv, fix_points=fix)
            gpu.solve()                                         # warm-up
            ts = []
            for _ in range(5):
                gpu.set_cameras(sc.t0, sc.rpy0)
                gpu.set_points(pts)
                t0 = time.perf_counter()
                g = gpu.solve()
                ts.append(time.perf_counter() - t0)
            gpu.close()
            rec = dict(gpu_ms=float(np.median(ts)) * 1e3, iterations=int(g["iterations"]),
                       final_cost=float(g["final_cost"]), setup_ms=setup_ms(sc, pts, fix))
            out["%s_%s" % (name, "motion_only" if fix else "full")] = rec
    # ten LM iterations of full BA at config 4 (the CPU oracle forms the dense reduced system there:
    # no CPU figure beside it)
    sc = synth.make_ba_config("c4", seed=0)
    gpu = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv)
    gpu.solve(max_num_iterations=2)
    ts = []
    for _ in range(3):
        gpu.set_cameras(sc.t0, sc.rpy0)
        gpu.set_points(sc.pts0)
        t0 = time.perf_counter()
        g = gpu.solve(max_num_iterations=10)
        ts.append(time.perf_counter() - t0)
    gpu.close()
    out["c4_full_10_iterations"] = dict(gpu_ms=float(np.median(ts)) * 1e3, iterations=int(g["iterations"]),
                                        successful_steps=int(g["successful_steps"]),
                                        initial_cost=float(g["initial_cost"]), final_cost=float(g["final_cost"]),
                                        setup_ms=setup_ms(sc, sc.pts0, False))
    out["adapter"] = bench_adapter()
    return out


def bench_adapter():
    """The drop-in path end to end: the C++ adapter with the reference's Optimizer interface
    (addPose / addKeys / addPoints / addReprojectionEdges / solve / getPose, a fresh Optimizer per
    solve as in Tracking::solveCamera, src/core/tracking.cc:79-120) on a 10-keyframe, 500-point
    window, timed inside tests/cpp/adapter_harness (median wall time of one whole sequence)."""
    import struct
    import subprocess
    import tempfile
    from visual_slam_amd import optimizer as vo, synth
    root = os.path.dirname(os.path.abspath(__file__))
    harness = os.path.join(root, "tests", "cpp", "adapter_harness")
    if not os.path.exists(harness):
        return dict(note="tests/cpp/adapter_harness not built")
    out = {}
    sc, tracks, keys = synth.make_tracks_scene(10, 500, seed=3)
    nf, npnt = tracks.shape
    poses = np.zeros((nf, 7), np.float32)
    for f in range(nf):
        poses[f, :3] = sc.t0[f]
        poses[f, 3:] = vo.rpy_to_quat_f32(sc.rpy0[f].astype(np.float32))
    with tempfile.TemporaryDirectory() as tmp:
        for fix in (1, 0):
            pw = (sc.pts_gt if fix else sc.pts0).astype(np.float32)
            blob = struct.pack("<4i", 0, nf, npnt, fix) + poses.tobytes()
            for f in range(nf):
                blob += struct.pack("<i", len(keys[f])) + keys[f].astype(np.int32).tobytes()
            blob += tracks.astype(np.int32).tobytes() + pw.tobytes()
            fi, fo = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
            with open(fi, "wb") as f:
                f.write(blob)
            r = subprocess.run([harness, fi, fo], capture_output=True, text=True, timeout=300,
                               env=dict(os.environ, VS_HARNESS_REPEAT="30"))
            ms = [float(l.split()[1]) for l in r.stdout.splitlines() if l.startswith("adapter_ms_median")]
            key = "motion_only" if fix else "full"
            out[key] = dict(adapter_ms=ms[0] if ms and r.returncode == 0 else None,
                            observations=int((tracks >= 0).sum()),
                            note="Optimizer built, solved and read back; 10 keyframes x 500 tracked points")
    return out


def bench_match(args, rank, world, dist, torch, n=4096, total_pairs=64):
    from visual_slam_amd import matcher as vm
    pairs = max(1, total_pairs // world)
    mt = vm.Matcher(pairs, n, n)
    mt.set_stream(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda")
    g.manual_seed(1234 + rank)
    keep = []
    for p in range(pairs):
        d0 = torch.randn(n, 256, device="cuda", generator=g)
        d0 = d0 / d0.norm(dim=1, keepdim=True)
        n_in = int(n * 0.7)
        d1 = torch.randn(n, 256, device="cuda", generator=g)
        src = torch.randperm(n, device="cuda", generator=g)[:n_in]
        d1[:n_in] = d0[src] + (0.3 / 16.0) * torch.randn(n_in, 256, device="cuda", generator=g)
        d1 = d1[torch.randperm(n, device="cuda", generator=g)]
        d1 = (d1 / d1.norm(dim=1, keepdim=True)).contiguous()
        keep.append((d0, d1))
    # the batch as the network would hand it over: one (pairs, n, 256) array per side
    D0 = torch.stack([k[0] for k in keep]).contiguous()
    D1 = torch.stack([k[1] for k in keep]).contiguous()
    keep = [(D0[p], D1[p]) for p in range(pairs)]
    side0, side1 = (D0.data_ptr(), pairs, n, n * 256), (D1.data_ptr(), pairs, n, n * 256)
    mt.set_pairs(0, side0, side1)
    torch.cuda.synchronize()
    steps = max(3, args.steps // 10)
    for _ in range(2):
        mt.run(pairs, 0.7)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        mt.run(pairs, 0.7)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    if world > 1:
        tt = torch.tensor([dt], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    per_batch = dt / steps
    out = dict(pairs=pairs * world, n0=n, n1=n, ms_per_batch=per_batch * 1e3,
               query_rows_per_s=pairs * world * n / per_batch, pairs_per_s=pairs * world / per_batch)
    if rank == 0:
        ms = mt.profile_run(pairs, 0.7, iters=3)
        flops = 2.0 * n * n * 256 * pairs
        ach = flops / (float(ms[0]) * 1e-3) / 1e12
        # the same batch with the ratio test on (BASELINE config 3: "ratio + mutual check")
        for _ in range(2):
            mt.run(pairs, 0.7, 0.8)
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        for _ in range(steps):
            mt.run(pairs, 0.7, 0.8)
        torch.cuda.synchronize()
        per_ratio = (time.perf_counter() - t4) / steps
        out["ratio_on"] = dict(ratio=0.8, ms_per_batch=per_ratio * 1e3, query_rows_per_s=pairs * n / per_ratio,
                               matches_pair0=len(mt.matches(0)[0]), slowdown_vs_ratio_off=per_ratio / per_batch)
        # handing the descriptors over (vs_matcher_set_pairs from device arrays: copy into the
        # handle, fp16 rounding and its error statistics) happens once per batch, before run
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        for _ in range(steps):
            mt.set_pairs(0, side0, side1)
            mt.run(pairs, 0.7)
        torch.cuda.synchronize()
        with_set = (time.perf_counter() - t5) / steps
        out["ms_per_batch_with_set_pairs"] = with_set * 1e3
        out["set_pairs_ms"] = (with_set - per_batch) * 1e3
        mt.run(pairs, 0.7)
        out["rescans_per_pair"] = mt.rescans() / pairs
        n_match = len(mt.matches(0)[0])
        out["matches_pair0"] = n_match
        out["stages_ms"] = dict(zip(["dot_top2", "merge", "exact+rescan", "compact"], [float(v) for v in ms[:4]]))
        out["roofline"] = dict(bound="mfma", achieved=ach, peak=MFMA_F16_PEAK_TFLOPS, unit="TFLOP/s",
                               frac=ach / MFMA_F16_PEAK_TFLOPS, traffic=None, kernel="dot_top2_kernel",
                               note="algorithmic flops 2*N0*N1*256 per pair = the flops the kernel issues "
                                    "(one fp16 MFMA term; exactness restored by certified rescans)")
        # what the matrix cores deliver on this chip with real data: a bare MFMA chain from registers
        # on random operands (the clock drops under the power limit), and on zeros for comparison
        import ctypes as C
        from visual_slam_amd import _lib
        lib = _lib.load()
        rate = C.c_float(0)
        if lib.vs_debug_mfma_rate(300, 1, C.byref(rate)) == 0:
            out["roofline"]["sustained_peak_measured"] = float(rate.value)
            out["roofline"]["frac_of_sustained_peak"] = ach / float(rate.value)
            out["roofline"]["sustained_peak_note"] = ("bare v_mfma_f32_32x32x16_f16 chain, operands in registers, "
                                                      "random data, measured in this run")
        if lib.vs_debug_mfma_rate(300, 0, C.byref(rate)) == 0:
            out["roofline"]["zero_operand_peak_measured"] = float(rate.value)
        # library GEMM of the same shape for scale: fp16 bmm through hipBLASLt, scores written to HBM
        a = torch.stack([k[0] for k in keep[:min(pairs, 8)]]).half()
        b = torch.stack([k[1] for k in keep[:min(pairs, 8)]]).half().transpose(1, 2).contiguous()
        for _ in range(2):
            torch.bmm(a, b)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(5):
            torch.bmm(a, b)
        torch.cuda.synchronize()
        lib_ms = (time.perf_counter() - t2) / 5 * 1e3 * (pairs / a.shape[0])
        out["library_gemm_fp16"] = dict(ms_per_batch=lib_ms, tflops=flops / (lib_ms * 1e-3) / 1e12,
                                        note="torch.bmm fp16 (hipBLASLt) of the same contraction, scores "
                                             "materialised, no top-2: a ceiling reference, not a matcher")
        if world == 1:
            # throughput with two batches in flight: a second handle on a second stream, batches
            # alternate, so one batch's merge / rescan / compaction kernels run beside the next
            # batch's score kernel.  Reported beside the one-stream number, not instead of it.
            s2 = torch.cuda.Stream()
            mt2 = vm.Matcher(pairs, n, n)
            mt2.set_stream(s2.cuda_stream)
            mt2.set_pairs(0, side0, side1)
            both = (mt, mt2)
            for it in range(4):
                both[it % 2].run(pairs, 0.7)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            for it in range(2 * steps):
                both[it % 2].run(pairs, 0.7)
            torch.cuda.synchronize()
            per2 = (time.perf_counter() - t3) / (2 * steps)
            out["two_streams"] = dict(ms_per_batch=per2 * 1e3, query_rows_per_s=pairs * n / per2,
                                      note="two handles on two streams, batches alternating")
            mt2.close()
    mt.close()
    return out


def bench_match_one_pair_sharded(args, rank, world, dist, torch, comm, n=4096):
    """ONE 4096 x 4096 frame pair split by query rows over the ranks (vs_matcher_run_sharded): every
    rank holds n / world rows of desc0 and all of desc1; the column nearest neighbours are combined
    by the library's minimum all-reduce of packed keys on the matcher's stream."""
    from visual_slam_amd import matcher as vm
    g = torch.Generator(device="cuda")
    g.manual_seed(4321)                                   # the same pair on every rank
    d0 = torch.randn(n, 256, device="cuda", generator=g)
    d0 = d0 / d0.norm(dim=1, keepdim=True)
    d1 = d0[torch.randperm(n, device="cuda", generator=g)] + (0.3 / 16.0) * torch.randn(n, 256, device="cuda", generator=g)
    d1 = (d1 / d1.norm(dim=1, keepdim=True)).contiguous()
    lo, hi = rank * n // world, (rank + 1) * n // world
    mine = d0[lo:hi].contiguous()
    mt = vm.Matcher(1, max(hi - lo, 1), n)
    mt.set_stream(torch.cuda.current_stream().cuda_stream)
    mt.set_pair(0, (mine.data_ptr(), hi - lo), (d1.data_ptr(), n))
    mt.set_comm(comm)
    steps = max(10, args.steps // 10)
    for _ in range(3):
        mt.run_sharded(1, lo, 0.7)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        mt.run_sharded(1, lo, 0.7)
    dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    per = float(tt.item()) / steps
    n_local = len(mt.matches(0)[0])
    cnt = torch.tensor([n_local], device="cuda", dtype=torch.int64)
    dist.all_reduce(cnt)
    mt.close()
    return dict(n0=n, n1=n, ranks=world, ms_per_pair=per * 1e3, query_rows_per_s=n / per, matches=int(cnt.item()),
                note="one pair, query rows split over the ranks, packed-key minimum all-reduce (32 KB) per pair")


def bench_tracking(torch):
    """Per-frame tracking helpers at frame scale (host buffers in, host buffers out)."""
    from visual_slam_amd import synth
    from visual_slam_amd import tracking as vt
    pose, K4, Pw, desc, key_xy, key_desc = synth.make_tracking_frame(2000, 4096, seed=0)
    d0, xy0, d1, xy1, K4s, bf, pose_s = synth.make_stereo_frame(4096, 4096, seed=0)
    out = {}
    for name, fn in (("search_projection_2000x4096", lambda: vt.search_projection(pose, K4, Pw, desc, key_xy, key_desc)),
                     ("compute_points3d_4096x4096", lambda: vt.compute_points3d(d0, xy0, d1, xy1, K4s, bf, pose_s))):
        fn()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        out[name + "_ms"] = (time.perf_counter() - t0) / 3 * 1e3
    p1, p2, K4t, k1, t1, k2, t2, xyz0 = synth.make_two_view_tracks(100000, seed=0)
    vt.triangulate_tracks(p1, p2, K4t, k1, t1, k2, t2, xyz0)
    t0 = time.perf_counter()
    for _ in range(3):
        vt.triangulate_tracks(p1, p2, K4t, k1, t1, k2, t2, xyz0)
    out["triangulate_tracks_100000_ms"] = (time.perf_counter() - t0) / 3 * 1e3
    out["note"] = "one-shot C ABI calls, PCIe copies of descriptors included"
    return out


def _median_time(fn, passes, warm=1):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(passes):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def _normal_equations_pass(orc, sc):
    """One LM linear step of the c2 shape the way the reference's solver takes it: the full damped
    normal equations over cameras and points, factored sparse (SuperLU in symmetric mode with a
    minimum-degree ordering stands in for CHOLMOD, which is not installed) and solved."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    p = orc.Problem(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv)
    t0 = time.perf_counter()
    _, res, Jc, Jp = p.evaluate()
    No, Nc, Np = sc.No, sc.Nc, sc.Np
    rows = np.repeat(np.arange(2 * No), 9)
    cols = np.concatenate([6 * sc.cam[:, None] + np.arange(6), 6 * Nc + 3 * sc.pt[:, None] + np.arange(3)], axis=1)
    cols = np.repeat(cols, 2, axis=0).reshape(-1)
    vals = np.concatenate([Jc.reshape(No, 2, 6), Jp.reshape(No, 2, 3)], axis=2).reshape(-1)
    J = sp.csr_matrix((vals, (rows, cols)), shape=(2 * No, 6 * Nc + 3 * Np))
    H = (J.T @ J + 1e-4 * sp.identity(J.shape[1])).tocsc()
    g = J.T @ res.reshape(-1)
    lu = spl.splu(H, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    lu.solve(-g)
    return time.perf_counter() - t0


def cpu_baseline(name):
    """The CPU oracle (fp64) on the same BA workload, residual + Jacobian + Schur: one thread (the
    reference runs its solver on one thread, src/core/tracking.cc:101-103) as the headline, the
    OpenMP variant on the host cores this process may use beside it; the sparse normal-equation
    route at the 50-camera shape; one full-size matcher pair on one thread and on all of them."""
    import ctypes as C
    from oracle import oracle as orc
    from visual_slam_amd import synth
    threads = max(1, min(orc.max_threads(), len(os.sched_getaffinity(0))))
    sc = synth.make_ba_config(name)
    p = orc.Problem(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv)
    n = 6 * sc.Nc
    dc = np.full(n, 1e-4); dp = np.full(3 * sc.Np, 1e-4)
    S = np.zeros((n, n)); rhs = np.zeros(n); Vinv = np.zeros((sc.Np, 9)); gp = np.zeros((sc.Np, 3))
    lib, d = orc.lib(), orc._d

    def one():
        lib.vo_schur(*p._args(), 0, d(dc), d(dp), d(S), d(rhs), None, None, d(Vinv), d(gp), None)

    def many():
        lib.vo_schur_mt(threads, *p._args(), d(dc), d(dp), d(S), d(rhs), d(Vinv), d(gp))

    passes = 10
    dt1 = _median_time(one, passes)
    dtn = _median_time(many, passes)
    what = "vo_schur (residual+Jacobian+dense-S Schur, fp64, gcc -O3) over the whole %s problem (%d observations)" % (name, sc.No)
    out = dict(value=sc.No / dt1, unit="observations/s", cores=1, kind="port",
               sample="median of %d passes of %s" % (passes, what))
    out["all_cores"] = dict(value=sc.No / dtn, unit="observations/s", cores=threads, kind="port",
                            sample="median of %d passes of the OpenMP variant of %s" % (passes, what))
    # the reference's own route (SPARSE_NORMAL_CHOLESKY, src/core/tracking.cc:103) beside the Schur
    # route, one linear step each at the 50-camera shape
    c2 = synth.make_ba_config("c2")
    dt_ne = _normal_equations_pass(orc, c2)
    p2 = orc.Problem(c2.K, c2.t0, c2.rpy0, c2.pts0, c2.cam, c2.pt, c2.uv)
    dt_sc = _median_time(lambda: p2.lm_step(1e4), passes)
    out["normal_equations"] = dict(value=c2.No / dt_ne, unit="observations/s", cores=1, kind="port", seconds=dt_ne,
                                   sample="one pass at c2 (%d observations): oracle residual+Jacobian, sparse J^T J "
                                          "over cameras and points, SuperLU symmetric-mode factorisation + solve "
                                          "(CHOLMOD is not installed)" % c2.No,
                                   schur_route=dict(value=c2.No / dt_sc, seconds=dt_sc,
                                                    sample="median of %d vo_lm_step passes at c2 (Schur + dense "
                                                           "Cholesky + back-substitution)" % passes))
    # whole LM solves at the reference's own operating point (config 1) and at config 2, one core
    out["solve"] = {}
    for nm in ("c1", "c2"):
        s_ = synth.make_ba_config(nm, seed=0)
        for fix in (True, False):
            pts = s_.pts_gt if fix else s_.pts0
            ref = orc.Problem(s_.K, s_.t0, s_.rpy0, pts, s_.cam, s_.pt, s_.uv, fix_points=fix)
            t0 = time.perf_counter()
            r = ref.solve()
            out["solve"]["%s_%s" % (nm, "motion_only" if fix else "full")] = dict(
                cpu_ms=(time.perf_counter() - t0) * 1e3, iterations=int(r["iterations"]), final_cost=float(r["final_cost"]),
                cores=1, kind="port", sample="one vo_solve run (LM loop of the oracle)")
    d0, d1 = synth.make_descriptors(4096, 4096, seed=0)
    dm1 = _median_time(lambda: orc.two_way_matching(d0, d1, 0.7), 3, warm=0)
    dmn = _median_time(lambda: orc.two_way_matching_mt(threads, d0, d1, 0.7), 5)
    out["matcher"] = dict(value=4096 / dm1, unit="query rows/s at 4096 candidates", cores=1, kind="port",
                          sample="median of 3 passes over one 4096x4096 pair (scalar loop, double accumulation)",
                          all_cores=dict(value=4096 / dmn, cores=threads,
                                         sample="median of 5 passes, rows spread over OpenMP threads"))
    return out


def spawn_ranks(n):
    """Starts n copies of this script, one rank per GPU, and returns the worst exit code.  Runs
    before anything in this process touches a GPU; rank 0's stdout (the JSON line) passes through."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    return max(abs(p.wait()) for p in procs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="c4", choices=["c1", "c2", "c4", "c5shard", "c5"])
    ap.add_argument("--no-extra", action="store_true", help="skip the matcher leg and larger BA shapes")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise a process group and all-reduce even with one rank (rehearsal of the N>1 path)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # rehearsal of the N>1 path on a box with fewer GPUs than ranks: VS_BENCH_BACKEND=gloo lets the
    # ranks share devices (RCCL refuses two ranks on one GPU); the driver's runs use nccl = RCCL
    backend = os.environ.get("VS_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dist = None
    comm = None
    if world > 1 or args.force_dist:
        import ctypes as C
        import torch.distributed as dist
        from visual_slam_amd import _lib
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
        # RCCL and gloo print banners on stdout when the communicator comes up: keep stdout for
        # the one JSON line by pointing file descriptor 1 at stderr until the first collective is done
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            warm = torch.zeros(8, device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(warm)
            torch.cuda.synchronize()
            if backend == "nccl":
                # the library's own communicator: rank 0 makes the id, torch.distributed carries it
                # (byte 128 of the message says whether rank 0 could make one).  Should the library's
                # communicator not come up on every rank, all ranks drop to torch's RCCL group together.
                lib = _lib.load()
                ident = torch.zeros(129, dtype=torch.uint8)
                if rank == 0:
                    buf = (C.c_char * 128)()
                    # VS_BENCH_NO_LIB_COMM=1 rehearses the fall-back
                    if not os.environ.get("VS_BENCH_NO_LIB_COMM") and lib.vs_comm_unique_id(buf) == 0:
                        ident[:128] = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8)
                        ident[128] = 1
                ident = ident.cuda()
                dist.broadcast(ident, 0)
                ident = ident.cpu()
                ok = int(ident[128])
                if ok:
                    raw = bytes(ident[:128].numpy().tobytes())
                    comm = C.c_void_p()
                    x = torch.ones(8, device="cuda", dtype=torch.float64)
                    ok = int(lib.vs_comm_create(C.byref(comm), rank, world, raw) == 0 and
                             lib.vs_comm_allreduce_sum(comm, C.c_void_p(x.data_ptr()), C.c_int64(8), None) == 0)
                    torch.cuda.synchronize()
                    ok = int(ok and float(x[0].item()) == float(world))
                flag = torch.tensor([ok], device="cuda", dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    trace("library communicator unavailable (%s): using torch.distributed's RCCL group"
                          % lib.vs_last_error_string().decode("utf-8", "replace"))
                    comm = None
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    trace("process group ready" if dist is not None else "single rank")
    ba = bench_ba(args, rank, world, dist, torch, comm)
    trace("BA leg done")
    extra = {}
    if not args.no_extra:
        extra["matcher"] = bench_match(args, rank, world, dist, torch)
        if comm is not None:
            extra["matcher_one_pair_sharded"] = bench_match_one_pair_sharded(args, rank, world, dist, torch, comm)
        if rank == 0 and world == 1:
            for nm in ("c2", "c4", "c5shard", "c5"):
                extra[nm] = bench_ba_shape(nm, torch)
            extra["tracking"] = bench_tracking(torch)
            extra["solve"] = bench_solve_small()
    if rank == 0:
        line = {
            "metric": "BA observations/sec (residual+J+Schur) & SuperPoint matches/sec, 1\u20138 GPU",
            "value": ba["value"], "unit": "observations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ba["ms_per_step"], "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d cameras, %d landmarks, %d observations%s" % (
                args.workload, ba["Nc"], ba["Np"], ba["No"],
                "" if world == 1 else " (landmarks sharded over %d GPUs, the reduced camera system combined every step: each rank all-gathers the rows its landmarks touch)" % world),
                "precision": "fp64 values, int32 indices"},
            "roofline": ba["roofline"],
        }
        if world > 1 or args.force_dist:
            via = "RCCL on the library's stream (vs_comm)" if comm is not None else "torch.distributed " + backend
            line["exchange"] = dict(ms_per_step=ba["exchange_ms_per_step"], send_bytes_per_rank=ba["exchange_send_bytes"],
                                    recv_bytes_per_rank=ba["exchange_recv_bytes"], via="vs_ba_exchange_reduced: all-gather of row segments, " + via,
                                    whole_buffer_allreduce=dict(ms_per_step=ba["allreduce_ms_per_step"], bytes=ba["allreduce_bytes"], via=via))
        if "two_streams" in ba:
            extra["ba_two_streams"] = ba["two_streams"]
        if "matcher" in extra:   # second half of the metric: SuperPoint query rows matched per second
            line["matcher_value"] = extra["matcher"]["query_rows_per_s"]
            line["matcher_unit"] = "query rows/s (64 pairs of 4096 x 4096, mutual check)"
        if not args.no_cpu and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.workload)
        elif not args.no_cpu:
            line["cpu_baseline"] = None
        line["extra"] = extra
        print(json.dumps(line))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        if comm is not None:
            from visual_slam_amd import _lib
            _lib.load().vs_comm_destroy(comm)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
