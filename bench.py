#!/usr/bin/env python3
"""bench.py — Mrays/s of the MI355X path tracer on the headline workload of BASELINE.json:
teapot_scene.yaml, 640x640, 100 spp (procedural stand-ins for the missing teapot.obj and
old_hall_4k.hdr), reference quirks, seed 0.

One "step" = one whole frame: path-trace kernel over this rank's interleaved row blocks, gather of the
fp32 linear film tiles (RCCL all_gather when N > 1), row permutation and Film resolve (tonemap -> u8)
on rank 0's copy.  Inputs (scene, BVH, env map) are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus N --steps K --warmup W          (no launcher: starts its own N ranks as a CHILD torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     — the path-trace kernel: algorithmic bytes per launch (SURVEY.md §8(d) prices, counted
                 exactly by the STATS build of the same kernel on the same seed) / average launch
                 duration measured with HIP events recorded on the launch stream around each launch.
  cpu_baseline — the CPU oracle (a port: the reference binary cannot be built) timed on
                 this box's host cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# peaks (HBM3E 8.0 TB/s spec; XCD-L2 gather 17 TB/s) live in hobbyraytracer_amd/benchline.py with the roofline arithmetic


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start N ranks as a CHILD process
    (python -m torch.distributed.run, one process per GPU) BEFORE this process imports torch or touches a GPU --
    never exec: replacing a process that has initialised the GPU takes the box down -- relay rank 0's JSON line
    and exit with the child's code."""
    import socket
    import subprocess
    from hobbyraytracer_amd.benchline import last_json_line
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)     # stderr passes through
    line = last_json_line(p.stdout)
    if line is not None:
        print(json.dumps(line), flush=True)
    else:
        sys.stderr.write(p.stdout[-4000:])
    if p.returncode != 0:
        raise SystemExit(p.returncode)
    if line is None:
        raise SystemExit("the ranks ended without printing a bench line")
    raise SystemExit(0)


def host_cores():
    """CPU threads this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    env = os.environ.get("HRT_CPU_THREADS")
    if env:
        n = max(1, int(env))
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=640)
    ap.add_argument("--spp", type=int, default=100)
    ap.add_argument("--scene", default="teapot_scene.yaml", help="file name under tests/golden/scenes")
    ap.add_argument("--quirks", choices=["reference", "fixed"], default="reference")
    ap.add_argument("--rows-per-block", type=int, default=8)
    ap.add_argument("--cpu-spp", type=int, default=12, help="spp of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--megakernel", action="store_true", help="time the persistent-lanes megakernel instead of the wavefront pipeline")
    ap.add_argument("--film-out", default="", help="rank 0 writes the gathered fp32 film of the last step to this .npy file (tests)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; 'gloo' only to rehearse N > 1 "
                    "with all ranks sharing one GPU, where RCCL refuses duplicate devices)")
    ap.add_argument("--force-dist", action="store_true", help="with one rank too: init_process_group(backend) + the all_gather of the film tiles "
                    "(so that the RCCL path has run on a one-GPU box before a multi-GPU node sees it)")
    ap.add_argument("--no-cli-wall-clock", action="store_true", help="skip the one run of the CLI binary behind config.cli_wall_clock_s")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus, sys.argv[1:])      # does not return

    import numpy as np
    import torch
    import torch.distributed as dist

    from hobbyraytracer_amd import api, benchline, tiles

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback of the product path)")
    if args.backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()     # rehearsal: ranks may share a device
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:   # --force-dist without a launcher: a one-rank group of our own
            os.environ.setdefault("MASTER_PORT", "29533"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    # ---- inputs: procedural assets + scene load + flatten + upload (all untimed)
    tmp = tempfile.mkdtemp(prefix=f"hrt_bench_r{rank}_")
    api.write_teapot_obj(os.path.join(tmp, "teapot.obj"), 1.0)
    api.write_hall_hdr(os.path.join(tmp, "old_hall_4k.hdr"), 4096, 2048)
    if "bust" in args.scene:
        api.write_bust_obj(os.path.join(tmp, "marble_bust_01.obj"), 1.0)   # ~100k triangles (config C5)
    # the loader prints the reference's console lines ("Loading scene: ...", scene.cpp) on stdout: keep stdout for the JSON line
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        hs = api.HostScene(os.path.join(ROOT, "tests", "golden", "scenes", args.scene), tmp)
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    W, H, spp = args.width, args.height, args.spp
    cam = hs.camera(W, H)
    quirks = api.QUIRKS_REFERENCE if args.quirks == "reference" else api.QUIRKS_FIXED
    params = api.default_params(W, H, spp, quirks=quirks, seed=0, megakernel=args.megakernel, timing=True)
    params_stats = api.default_params(W, H, spp, quirks=quirks, seed=0, stats=True, megakernel=args.megakernel)
    dev = api.DeviceScene(hs.flat_ptr, local_rank)

    R = args.rows_per_block
    layout = tiles.StripeLayout(H, W, R, world)
    my_rows = layout.rows[rank]
    tile = torch.zeros((layout.max_rows, W, 3), dtype=torch.float32, device="cuda")
    gathered = torch.zeros((world * layout.max_rows, W, 3), dtype=torch.float32, device="cuda") if use_dist else None
    film_lin = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    film_u8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step(p):
        dev.render_stripes_device(cam, p, R, rank, world, tile.data_ptr(), stream)
        tiles.gather_film(tile, layout, dist if use_dist else None, out=film_lin, gathered=gathered, force_collective=args.force_dist)   # RCCL all_gather
        if rank == 0:
            dev.resolve_u8_device(film_lin.data_ptr(), W * H, film_u8.data_ptr(), stream)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- counting launch (untimed): exact algorithmic bytes of one launch of this rank
    step(params_stats)
    sync()
    st_count = dev.stats()
    alg_bytes_launch = st_count.algorithmic_bytes(my_rows * W)

    for _ in range(args.warmup):
        step(params)
    sync()
    dev.stats()  # clear counters / kernel timers

    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(params)
    sync()
    t1 = time.perf_counter()
    st = dev.stats()

    if rank == 0 and args.film_out:
        np.save(args.film_out, film_lin.cpu().numpy())

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
    counts = torch.tensor([float(st.rays), float(st.samples), float(alg_bytes_launch)], dtype=torch.float64, device="cuda")
    kern = torch.tensor([st.kernel_ms / max(1, st.launches)], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        kern_max = kern.clone()
        dist.all_reduce(kern_max, op=dist.ReduceOp.MAX)
    else:
        kern_max = kern
    seconds = float(elapsed.item())
    total_rays, total_samples, total_bytes_launch = [float(x) for x in counts.tolist()]
    kernel_ms = float(kern_max.item())

    if rank == 0:
        ms_per_step = seconds / args.steps * 1e3
        mrays = total_rays / seconds / 1e6
        # roofline of the dominant kernel on rank 0's device.
        #   wavefront pipeline: k_wf_ext (BVH traversal): bytes = 32 B per box + 36 B per triangle it tested,
        #   launches = its launches per frame (one per round per mesh), time = HIP events around each launch.
        #   megakernel: k_pathtrace, all of SURVEY §8(d)'s bytes, one launch per frame.
        frame_ms = float(kern.item())
        roof_note = None
        if args.megakernel or st.traversal_launches == 0:
            kname, k_launches = ("k_pathtrace" if args.megakernel else "k_wf_gen+k_wf_shade+k_wf_reduce (whole frame)"), 1
            k_bytes, k_ms = float(alg_bytes_launch), frame_ms
            if not args.megakernel:   # a scene without a mesh has no BVH traversal: fp32 VALU / divergence bound
                roof_note = "no mesh in this scene: the frame moves almost no memory, an HBM fraction is not meaningful (VALU / divergence bound)"
        else:
            kname, k_launches = "k_wf_ext", st.traversal_launches / max(1, st.launches)
            # bytes of exactly those launches (rounds a small batch runs inside k_wf_tail are neither timed nor counted here)
            k_bytes = (32.0 * st_count.traversal_box_tests + 36.0 * st_count.traversal_tri_tests) / k_launches
            k_ms = st.traversal_ms / st.traversal_launches
        # HBM traffic and issue-side counters of that kernel cannot be measured inside this process (PMC counters need their own
        # rocprofv3 passes): the figures of the newest committed profile of the same workload, if any (profiles/collect.sh)
        traffic = issue = None
        if W == 640 and H == 640 and spp == 100 and args.scene == "teapot_scene.yaml" and args.quirks == "reference" and world == 1:
            traffic, issue, _ = benchline.newest_profile_figures(ROOT, kname)
        trav = kname == "k_wf_ext"
        roofline = benchline.roofline_block(
            kname, k_bytes, k_ms, k_launches, frame_ms, alg_bytes_launch,
            st_count.box_tests / max(1, st_count.rays), st_count.tri_tests / max(1, st_count.rays),
            trav_box_tests=st_count.traversal_box_tests if trav else None, trav_tri_tests=st_count.traversal_tri_tests if trav else None,
            traffic=traffic, issue=issue, note=roof_note)
        result = {
            "metric": f"Mrays/sec (path segments/s), {args.scene} {W}x{H} {spp}spp",
            "value": round(mrays, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (procedural teapot.obj ~6.2k tris" + (" / marble_bust_01.obj ~100k tris" if "bust" in args.scene else "") +
                    " + procedural 4096x2048 old_hall_4k.hdr stand-ins; seed 0)",
            "config": {"workload": f"{args.scene} {W}x{H} {spp}spp, quirks={args.quirks}, max_depth=50",
                       "render_path": "megakernel k_pathtrace" if args.megakernel else "wavefront pipeline k_wf_gen/pre/ext/shade/reduce",
                       "parallelism": f"image row blocks of {R} rows interleaved over {world} GPU(s); RCCL all_gather of fp32 film tiles" if world > 1 else "single GPU",
                       "rays_per_step": total_rays / args.steps, "samples_per_step": total_samples / args.steps,
                       "msamples_per_s": round(total_samples / seconds / 1e6, 3),
                       "wall_clock_s_per_frame": round(seconds / args.steps, 6),
                       "reference_readme_wall_clock_s": 150.0},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline and args.cpu_spp > 0:
            from oracle import oracle_py as orc
            cores = host_cores()
            cp = api.default_params(W, H, args.cpu_spp, quirks=quirks, seed=0)
            wd = orc.World(hs.flat_ptr)
            c0 = time.perf_counter()
            _, cst = wd.render_tile(cam, cp, threads=cores)
            c1 = time.perf_counter()
            cpu_mrays = cst.rays / (c1 - c0) / 1e6
            result["cpu_baseline"] = {"value": round(cpu_mrays, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
                                      "sample": f"same scene and film {W}x{H} at {args.cpu_spp} spp (of {spp}), {cst.rays} segments in {c1 - c0:.2f} s, std::thread over rows"}
            result["config"]["gpu_over_cpu"] = round(mrays / cpu_mrays, 2)      # (a reported baseline, not the target)
        if world == 1 and not args.no_cli_wall_clock and not args.megakernel:
            # the reference's own stopwatch spans the whole process (main.cpp:144,184: load + render + image file): one run of the CLI
            # binary on the same workload, untimed by the loop above
            import subprocess
            from hobbyraytracer_amd import api as _api
            try:
                cp = subprocess.run([_api.CLI_PATH, os.path.join(ROOT, "tests", "golden", "scenes", args.scene), "--assets", tmp, "--size", f"{W}x{H}",
                                     "--spp", str(spp), "--quirks", args.quirks, "--out", os.path.join(tmp, "cli.png"), "--stats"],
                                    capture_output=True, text=True, timeout=600)
                line = [l for l in cp.stdout.splitlines() if l.startswith("{")][-1]
                cj = json.loads(line)
                result["config"]["cli_wall_clock_s"] = cj["wall_s"]
                result["config"]["cli_load_s"] = cj["load_s"]
                result["config"]["cli_render_s"] = cj["render_s"]
            except Exception as e:   # the number is an extra: never lose the bench line over it
                result["config"]["cli_wall_clock_s"] = None
                result["config"]["cli_wall_clock_error"] = str(e)[:200]
        print(json.dumps(result), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
