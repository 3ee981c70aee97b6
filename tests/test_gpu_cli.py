"""The CLI (hobbyraytracer_amd/bin/hobbyraytracer), drop-in for the reference executable (main.cpp:142-195):
positional scene argument, default scene name in the cwd, console lines, exit code = Film::outputFilm()'s int
(1 on success, Q-12), PNG equal to the oracle's film."""
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cli_renders_the_default_scene_and_matches_the_oracle(built, assets, scenes_dir, tmp_path):
    from hobbyraytracer_amd import api
    from oracle import oracle_py as orc
    # the reference resolves everything against the cwd (scene.cpp:294-296, mesh.cpp:56): stage a cwd like a user's
    for f in ("teapot.obj", "old_hall_4k.hdr"):
        shutil.copy(os.path.join(assets, f), tmp_path / f)
    shutil.copy(os.path.join(scenes_dir, "teapot_scene.yaml"), tmp_path / "teapot_scene.yaml")
    p = subprocess.run([api.CLI_PATH, "--size", "64x64", "--spp", "6", "--stats"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert p.returncode == 1, p.stderr                       # main.cpp:194 returns stb's 1 on success
    out = p.stdout
    assert "Loading scene: teapot_scene.yaml" in out and "Loaded mesh: teapot.obj" in out and "Indexed file: teapot.obj" in out
    assert "Loaded scene: teapot_scene.yaml! (completed in" in out and "Pixels rendered: 4096/4096" in out and "Done! (completed in" in out
    img = api.read_png(str(tmp_path / "teapot.png"))          # film.output of the YAML file
    hs = api.HostScene(str(tmp_path / "teapot_scene.yaml"), str(tmp_path))
    ref, _ = orc.World(hs.flat_ptr).render_tile(hs.camera(64, 64), api.default_params(64, 64, 6))
    assert np.array_equal(img, orc.resolve_u8(ref))
    # explicit scene path + fixed quirks + BMP fallback for an unknown suffix (film.cpp:73-78)
    p = subprocess.run([api.CLI_PATH, "teapot_scene.yaml", "--size", "48x32", "--spp", "2", "--quirks", "fixed", "--out", "x.foo", "--seed", "7"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert p.returncode == 1 and "File type not supported, generating bitmap!" in p.stdout
    assert open(tmp_path / "x.foo", "rb").read(2) == b"BM"
    # --dump-linear: the fp32 film, bit for bit the oracle's
    p = subprocess.run([api.CLI_PATH, "--size", "64x64", "--spp", "6", "--dump-linear", "film.pfm"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert p.returncode == 1
    assert np.array_equal(api.read_pfm(str(tmp_path / "film.pfm")).view(np.uint32), ref.view(np.uint32))


def test_cli_make_assets(built, tmp_path):
    from hobbyraytracer_amd import api
    p = subprocess.run([api.CLI_PATH, "--make-assets", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    for f in ("teapot.obj", "marble_bust_01.obj", "old_hall_4k.hdr"):
        assert os.path.getsize(tmp_path / f) > 1000
    env = api.read_hdr(str(tmp_path / "old_hall_4k.hdr"))
    assert env.shape == (2048, 4096, 3) and 49 < env.max() <= 50.5     # windows "up to ~50.0" 


def test_cli_progressive_checkpoint_resume(built, assets, scenes_dir, tmp_path):
    """--progressive / --checkpoint / --max-passes / --resume: a render interrupted after one pass and continued
    from its checkpoint writes the same PNG, byte for byte in pixels, as the uninterrupted one-shot render; the
    image on disk between passes is the preview of the samples so far; a checkpoint of another render is refused."""
    from hobbyraytracer_amd import api
    for f in ("teapot.obj", "old_hall_4k.hdr"):
        shutil.copy(os.path.join(assets, f), tmp_path / f)
    shutil.copy(os.path.join(scenes_dir, "shiny_teapot.yaml"), tmp_path / "s.yaml")
    common = ["s.yaml", "--size", "80x45", "--spp", "12", "--seed", "3"]

    def run(*extra):
        return subprocess.run([api.CLI_PATH, *common, *extra], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    p = run("--out", "one.png")
    assert p.returncode == 1, p.stderr
    p = run("--out", "two.png", "--progressive", "5", "--checkpoint", "ck.bin", "--max-passes", "1")
    assert p.returncode == 1 and "Samples rendered: 5/12" in p.stdout
    preview = api.read_png(str(tmp_path / "two.png"))
    p5 = subprocess.run([api.CLI_PATH, "s.yaml", "--size", "80x45", "--spp", "5", "--seed", "3", "--out", "five.png"], cwd=tmp_path,
                        capture_output=True, text=True, timeout=600)
    assert p5.returncode == 1 and np.array_equal(preview, api.read_png(str(tmp_path / "five.png")))
    p = run("--out", "two.png", "--progressive", "4", "--checkpoint", "ck.bin", "--resume")      # 5 + 4 + 3
    assert p.returncode == 1 and "Resumed at sample 5/12" in p.stdout and "Samples rendered: 9/12" in p.stdout
    assert np.array_equal(api.read_png(str(tmp_path / "two.png")), api.read_png(str(tmp_path / "one.png")))
    p = subprocess.run([api.CLI_PATH, "s.yaml", "--size", "80x45", "--spp", "12", "--seed", "4", "--checkpoint", "ck.bin", "--resume"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert p.returncode != 1 and "different render" in p.stderr


def test_multi_gpu_session_through_rccl_on_one_device(built, assets, scenes_dir):
    """hrt_multi_* (the C++ host's multi-GPU path): with force_rccl the session owns an RCCL communicator (ncclCommInitAll over
    its one device here) and gathers the device-resident stripes with ncclAllGather before the first device puts the rows in
    film order and resolves them -- film, u8 film and segment count equal hrt_render_tile's / hrt_resolve_u8's, in one shot and
    in passes continued from a checkpoint's sums; without it the gather is skipped and the result is the same."""
    from hobbyraytracer_amd import api
    hs = api.HostScene(f"{scenes_dir}/teapot_scene.yaml", assets)
    W, H, spp = 100, 77, 6            # 77 rows: ten blocks of 8 rows, the last one short
    cam = hs.camera(W, H)
    p = api.default_params(W, H, spp, stats=True)
    dev = api.DeviceScene(hs.flat_ptr, 0)
    ref, sref = dev.render_tile(cam, p)
    ref8 = dev.resolve_u8(ref)
    for force in (True, False):
        m = api.MultiScene(hs.flat_ptr, (0,), force_rccl=force)
        assert m.uses_rccl == force
        img, u8, st = m.render(cam, p)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and np.array_equal(u8, ref8) and st.rays == sref.rays
        # progressive: 2 + 4 samples, the second pass continued from the first pass's sums as a checkpoint would
        sums, prev8, _ = m.render(cam, p, sample_first=0, sample_count=2)
        two, _ = dev.render_tile(cam, api.default_params(W, H, 2))
        assert np.array_equal(prev8, dev.resolve_u8(two))                  # the preview is the 2-sample film
        m2 = api.MultiScene(hs.flat_ptr, (0,), force_rccl=force)
        img2, u82, _ = m2.render(cam, p, sample_first=2, sample_count=-1, resume_sums=sums)
        assert np.array_equal(img2.view(np.uint32), ref.view(np.uint32)) and np.array_equal(u82, ref8)
        m.close(); m2.close()
    dev.close()


def test_multi_gpu_session_with_several_ranks_in_loopback(built, assets, scenes_dir):
    """The threaded G > 1 session of hrt_multi_render on the box's ONE device (hrt_multi_create's loopback mode: logical ranks share
    the device, the gather is one device copy per rank instead of ncclAllGather; everything else is the production code): one host
    thread + stream + scene per rank, padded shares with H not a multiple of R x G, k_unstripe with G > 1, idle ranks (48 rows = 6
    blocks for 8 ranks), and a checkpoint written by G = 3 continued with G = 2 and G = 8 through k_restripe -- films, u8 films and
    segment counts equal hrt_render_tile's bit for bit.  What this does NOT run: ncclCommInitAll / ncclAllGather with more than one
    device."""
    from hobbyraytracer_amd import api
    hs = api.HostScene(f"{scenes_dir}/teapot_scene.yaml", assets)
    dev = api.DeviceScene(hs.flat_ptr, 0)
    for (W, H, spp, Gs, R) in ((100, 77, 6, (2, 3, 8), 8), (48, 48, 5, (8,), 8), (64, 50, 4, (3, 5), 4)):
        cam = hs.camera(W, H)
        p = api.default_params(W, H, spp, stats=True)
        ref, sref = dev.render_tile(cam, p)
        ref8 = dev.resolve_u8(ref)
        for G in Gs:
            m = api.MultiScene(hs.flat_ptr, (0,) * G, loopback=True)
            assert not m.uses_rccl
            img, u8, st = m.render(cam, p, rows_per_block=R)
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (W, H, G)
            assert np.array_equal(u8, ref8) and st.rays == sref.rays and st.samples == W * H * spp
            # a second frame on the same session (buffers reused, sums restarted at sample 0)
            img_b, _, _ = m.render(cam, p, rows_per_block=R, want_u8=False)
            assert np.array_equal(img_b.view(np.uint32), ref.view(np.uint32))
            m.close()
    # checkpoint across rank counts: 2 samples with G = 3, the rest with G = 2 and with G = 8 (restripe of a film-order checkpoint)
    W, H, spp = 100, 77, 6
    cam = hs.camera(W, H)
    p = api.default_params(W, H, spp)
    ref, _ = dev.render_tile(cam, p)
    two, _ = dev.render_tile(cam, api.default_params(W, H, 2))
    m3 = api.MultiScene(hs.flat_ptr, (0, 0, 0), loopback=True)
    sums, prev8, _ = m3.render(cam, p, sample_first=0, sample_count=2)
    assert np.array_equal(prev8, dev.resolve_u8(two))
    # ... continued on the same session in two more passes
    m3.render(cam, p, sample_first=2, sample_count=3, want_u8=False)
    img3, _, _ = m3.render(cam, p, sample_first=5, sample_count=-1)
    assert np.array_equal(img3.view(np.uint32), ref.view(np.uint32))
    m3.close()
    for G in (2, 8):
        m = api.MultiScene(hs.flat_ptr, (0,) * G, loopback=True)
        img, u8, _ = m.render(cam, p, sample_first=2, sample_count=-1, resume_sums=sums)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), G
        assert np.array_equal(u8, dev.resolve_u8(ref))
        m.close()
    # a device listed twice stays an error outside the test mode
    with pytest.raises(Exception, match="listed twice"):
        api.MultiScene(hs.flat_ptr, (0, 0))
    dev.close()


def test_cli_rccl_flag_gives_the_same_image(built, assets, scenes_dir, tmp_path):
    """The CLI's --gpus path is the multi-GPU session: `--rccl` sends the one-device film through the RCCL gather too."""
    from hobbyraytracer_amd import api
    import shutil
    for f in ("teapot.obj", "old_hall_4k.hdr"):
        shutil.copy(os.path.join(assets, f), tmp_path / f)
    shutil.copy(f"{scenes_dir}/teapot_scene.yaml", tmp_path / "s.yaml")
    outs = []
    for extra in ([], ["--rccl"]):
        name = "b.png" if extra else "a.png"
        p = subprocess.run([api.CLI_PATH, "s.yaml", "--size", "64x48", "--spp", "4", "--out", name, "--stats", *extra], cwd=tmp_path,
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 1, p.stderr
        assert '"wall_s"' in p.stdout and '"load_s"' in p.stdout          # the reference's whole-process stopwatch (main.cpp:144,184)
        outs.append(api.read_png(str(tmp_path / name)))
    assert np.array_equal(outs[0], outs[1])


def test_cli_bvh_lbvh_gives_the_same_image_with_fixed_quirks(built, assets, scenes_dir, tmp_path):
    """`--bvh lbvh`: the meshes' culling trees come from the GPU builder (hrt_bvh_build_device).  The film does not depend on the
    culling tree (fixed quirks: under the reference's, self-hit winners follow the reference tree over the soup in leaf order)."""
    from hobbyraytracer_amd import api
    import shutil
    for f in ("teapot.obj", "old_hall_4k.hdr"):
        shutil.copy(os.path.join(assets, f), tmp_path / f)
    shutil.copy(f"{scenes_dir}/teapot_scene.yaml", tmp_path / "s.yaml")
    outs = []
    for extra in (["--bvh", "sah"], ["--bvh", "lbvh"]):
        name = extra[1] + ".png"
        p = subprocess.run([api.CLI_PATH, "s.yaml", "--size", "64x48", "--spp", "4", "--quirks", "fixed", "--out", name, *extra], cwd=tmp_path,
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 1, p.stderr
        outs.append(api.read_png(str(tmp_path / name)))
    assert np.array_equal(outs[0], outs[1])
    p = subprocess.run([api.CLI_PATH, "s.yaml", "--bvh", "octree"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert p.returncode == 255 and "sah, gpu-sah or lbvh" in p.stderr


def test_bench_under_torchrun_two_ranks_on_one_gpu(built, assets, scenes_dir, tmp_path):
    """bench.py exactly as the driver launches it for N > 1 (python -m torch.distributed.run, one process per rank,
    rendezvous on 127.0.0.1), here with 2 ranks sharing the one GPU of the box and gloo as the collective backend (RCCL
    refuses two ranks on one device): interleaved row blocks + all_gather + row permutation give the single-process film
    bit for bit, and the JSON line carries the contract's fields."""
    import json
    import sys
    from hobbyraytracer_amd import api
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "1", "--warmup", "0", "--width", "96", "--height", "72", "--spp", "4", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *common, "--film-out", str(tmp_path / "one.npy")],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29517", os.path.join(root, "bench.py"), "--gpus", "2", *common, "--backend", "gloo",
                          "--film-out", str(tmp_path / "two.npy")], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    a, b = np.load(tmp_path / "one.npy"), np.load(tmp_path / "two.npy")
    assert a.shape == (72, 96, 3) and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    lines = [l for l in two.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 prints ONE JSON line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 1 and j["unit"] == "Mrays/s" and j["scaling"] == "strong" and j["value"] > 0
    assert j["config"]["samples_per_step"] == 96 * 72 * 4 and "roofline" in j and "cpu_baseline" not in j   # cpu_baseline: N = 1 only
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert j1["config"]["rays_per_step"] == j["config"]["rays_per_step"]      # both ranks' segments add up to the single-GPU count


def test_bench_one_rank_through_the_nccl_backend(built, tmp_path):
    """bench.py --force-dist: with ONE rank the bench still runs init_process_group("nccl") (= RCCL on ROCm) and the
    all_gather_into_tensor of the film tiles -- the collective path of the multi-GPU bench has then run on an MI355X before a
    multi-GPU node sees it -- and its line equals the plain one: same film, same segment count."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "1", "--warmup", "0", "--width", "96", "--height", "72", "--spp", "4", "--no-cpu-baseline", "--no-cli-wall-clock"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    outs = []
    for extra, name in (([], "a.npy"), (["--force-dist", "--backend", "nccl"], "b.npy")):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *common, *extra, "--film-out", str(tmp_path / name)],
                           capture_output=True, text=True, timeout=900, cwd=root, env=env)
        assert p.returncode == 0, p.stderr[-3000:]
        outs.append(json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0]))
    a, b = np.load(tmp_path / "a.npy"), np.load(tmp_path / "b.npy")
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert outs[0]["config"]["rays_per_step"] == outs[1]["config"]["rays_per_step"] and outs[1]["n_gpus"] == 1


def test_bench_default_workload_prints_the_whole_line(built):
    """bench.py at its DEFAULT workload (the headline frame: no --width / --height / --spp), one step: rc 0, exactly one JSON line,
    with roofline (incl. the committed-profile branch: traffic, hbm_measured_frac, issue, l2_frac), cpu_baseline and the CLI's wall
    clock.  Round 2's driver bench died in exactly the branch that only this workload takes."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=900, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 1 and j["value"] > 1000 and j["dtype"] == "f32"
    assert "640x640 100spp" in j["config"]["workload"] and j["config"]["samples_per_step"] == 640 * 640 * 100
    assert j["config"]["cli_wall_clock_s"] and j["config"]["cli_wall_clock_s"] > 0
    roof = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "hbm_measured_frac", "issue", "l2_frac", "kernel_ms_per_launch"):
        assert k in roof, k
    assert roof["kernel"] == "k_wf_ext" and roof["traffic"] > 0 and 0 < roof["l2_frac"] < 1 and roof["peak"] == 8000.0
    assert abs(roof["achieved"] / roof["peak"] - roof["frac"]) < 1e-3
    assert roof["kernel_ms_per_frame"] < j["ms_per_step"]          # the kernel is part of the frame
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "Mrays/s"


def test_bench_gpus_2_without_a_launcher(built, tmp_path):
    """`python bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset), the way the driver starts N = 1: the parent starts
    its own ranks as a child torch.distributed.run and relays rank 0's ONE line.  (Two ranks share the box's one GPU, so gloo.)"""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    common = ["--steps", "1", "--warmup", "0", "--width", "96", "--height", "72", "--spp", "4", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *common, "--film-out", str(tmp_path / "one.npy")],
                         capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", *common, "--film-out", str(tmp_path / "two.npy")],
                         capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    lines = [l for l in two.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, two.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and "roofline" in j
    a, b = np.load(tmp_path / "one.npy"), np.load(tmp_path / "two.npy")
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_progress_counter_and_the_reporter_thread(built, assets, scenes_dir, tmp_path):
    """main.cpp:95-109: `pixelsCompleted` + the reporter thread.  Here: HRT_FLAG_PROGRESS keeps a host-mapped counter of ended
    camera paths up to date (one tiny launch per round), hrt_scene_progress / hrt_multi_progress read it from another thread while
    the render runs, and the CLI prints "Pixels rendered: x/N" from it.  The film is the same with and without the flag."""
    import threading
    import time
    from hobbyraytracer_amd import api
    hs = api.HostScene(f"{scenes_dir}/teapot_scene.yaml", assets)
    W, H, spp = 384, 384, 100             # ~20 ms of rendering: the poller below sees a dozen intermediate values
    cam = hs.camera(W, H)
    dev = api.DeviceScene(hs.flat_ptr, 0)
    assert dev.progress() == (0, 0)
    ref, _ = dev.render_tile(cam, api.default_params(W, H, spp))
    seen = []
    stop = threading.Event()

    def poll():
        while not stop.is_set():
            seen.append(dev.progress())
            time.sleep(0.0005)
    th = threading.Thread(target=poll)
    th.start()
    img, _ = dev.render_tile(cam, api.default_params(W, H, spp, progress=True))
    stop.set(); th.join()
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    total = W * H * spp
    assert dev.progress() == (total, total)
    done = [d for d, t in seen if t == total]
    assert done == sorted(done) and all(0 <= d <= total for d in done)          # monotonic, bounded
    assert any(0 < d < total for d in done), "no intermediate value was seen while the render ran"
    # several batches (a small slot cap): the counter goes on across them and still ends at the total
    os.environ["HRT_WF_MAX_SLOTS"] = str(W * H * 8)
    try:
        dev2 = api.DeviceScene(hs.flat_ptr, 0)
        img2, _ = dev2.render_tile(cam, api.default_params(W, H, spp, progress=True))
        assert np.array_equal(img2.view(np.uint32), ref.view(np.uint32)) and dev2.progress() == (total, total)
        dev2.close()
    finally:
        del os.environ["HRT_WF_MAX_SLOTS"]
    # megakernel path: 0 until the launch is over, then everything
    img3, _ = dev.render_tile(cam, api.default_params(W, H, 4, progress=True, megakernel=True))
    assert dev.progress() == (W * H * 4, W * H * 4)
    # the session sums its ranks (loopback: 3 logical ranks on the one device)
    m = api.MultiScene(hs.flat_ptr, (0, 0, 0), loopback=True)
    m.render(cam, api.default_params(W, H, 8, progress=True))
    assert m.progress() == (W * H * 8, W * H * 8)
    m.close(); dev.close()
    # the CLI's reporter: intermediate lines at a 1 ms interval, the reference's first and last lines always
    for f in ("teapot.obj", "old_hall_4k.hdr"):
        shutil.copy(os.path.join(assets, f), tmp_path / f)
    shutil.copy(os.path.join(scenes_dir, "teapot_scene.yaml"), tmp_path / "teapot_scene.yaml")
    p = subprocess.run([api.CLI_PATH, "--size", "512x512", "--spp", "128", "--progress-ms", "1"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert p.returncode == 1, p.stderr
    import re
    vals = [int(x) for x in re.findall(r"Pixels rendered: (\d+)/262144", p.stdout)]
    assert vals[0] == 0 and vals[-1] == 262144 and vals == sorted(vals) and len(set(vals)) > 2, vals[:20]
    q = subprocess.run([api.CLI_PATH, "--size", "64x64", "--spp", "4", "--no-progress"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert q.returncode == 1 and re.findall(r"Pixels rendered: (\d+)/4096", q.stdout) == ["0", "4096"]
