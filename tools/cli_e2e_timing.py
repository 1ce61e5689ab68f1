"""Where does the wall time of `dashing-amd dist` over BASELINE configs[1] (1 000 x 5 Mbp FASTA on a RAM-backed file system)
go?  Runs the CLI with DSH_TIMING=1 (phase times on stderr) with the parse on the device (default) and on the host
(DSH_HOST_PARSE=1), several thread counts; prints one JSON line per run with the [timing] lines."""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    G, L = int(os.environ.get("G", "1000")), int(os.environ.get("L", "5000000"))
    dev = torch.device("cuda", 0)
    d = tempfile.mkdtemp(prefix="dsh_e2e_", dir="/dev/shm")
    try:
        lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
        g = torch.Generator(device=dev)
        g.manual_seed(7)
        paths = []
        root = torch.randint(0, 4, (L,), generator=g, device=dev, dtype=torch.uint8)
        for i in range(G):
            mut = torch.rand(L, generator=g, device=dev) < 0.02
            codes = torch.where(mut, torch.randint(0, 4, (L,), generator=g, device=dev, dtype=torch.uint8), root)
            body = torch.cat([lut[codes.long()].view(-1, 80), torch.full((L // 80, 1), 10, dtype=torch.uint8, device=dev)], dim=1).view(-1)
            pth = os.path.join(d, "g%04d.fna" % i)
            with open(pth, "wb") as f:
                f.write(b">genome%d\n" % i)
                f.write(body.cpu().numpy().tobytes())
            paths.append(pth)
        lst = os.path.join(d, "paths.txt")
        open(lst, "w").write("\n".join(paths) + "\n")
        del root
        torch.cuda.empty_cache()
        cli = os.path.join(ROOT, "dashing_amd", "dashing-amd")
        out = os.path.join(d, "dist.bin")
        ref = None
        if os.environ.get("ROCPROF_OUT"):
            # the same command under rocprofv3 --kernel-trace --stats: what the device does with the 5 GB (decode kernels,
            # k_sketch, the dist kernels), per kernel
            import glob

            td = tempfile.mkdtemp(prefix="kt_", dir="/tmp")
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "-d", td, "-o", "p", "--output-format", "csv", "--", cli, "dist", "-k", "31", "-S", "10",
                                "-p", "16", "-b", "--avoid-sorting", "-O", out, "-o", os.devnull, "-F", lst], capture_output=True, cwd="/tmp", timeout=900,
                               env=dict(os.environ, DSH_FULL_TEARDOWN="1"))  # (the profiler writes its files at exit: no _Exit)
            stats = glob.glob(os.path.join(td, "**", "*kernel_stats.csv"), recursive=True)
            if stats:
                shutil.copy(stats[0], os.environ["ROCPROF_OUT"])
            print(json.dumps({"what": "CLI under rocprofv3 --kernel-trace --stats", "rc": r.returncode, "stats_file": os.environ["ROCPROF_OUT"] if stats else None,
                              "stderr_tail": r.stderr.decode(errors="replace")[-300:] if r.returncode else ""}), flush=True)
            shutil.rmtree(td, ignore_errors=True)
            if os.environ.get("ROCPROF_ONLY"):
                return
        for _ in range(3):  # what a process that does nothing costs: exec + dynamic loading of the HIP runtime + exit
            t0 = time.perf_counter()
            subprocess.run([cli, "--help"], capture_output=True)
            print(json.dumps({"what": "dashing-amd --help (exec + ld.so + exit)", "wall_s": round(time.perf_counter() - t0, 4)}), flush=True)
        runs = (({}, 16), ({"DSH_HOST_PARSE": "1"}, 16), ({}, 16), ({"DSH_HOST_PARSE": "1"}, 16))
        if os.environ.get("SHORT"):
            runs = (({}, 16), ({"DSH_HOST_PARSE": "1"}, 16), ({"DSH_FULL_TEARDOWN": "1"}, 16), ({"DSH_HOST_PARSE": "1", "DSH_FULL_TEARDOWN": "1"}, 16)) * 3
        for th in (2, 4, 8) if not os.environ.get("SHORT") else ():
            runs += (({}, th), ({"DSH_HOST_PARSE": "1"}, th))
        if not os.environ.get("SHORT"):
            runs += (({}, 16), ({"DSH_HOST_PARSE": "1"}, 16))
        for env_extra, threads in runs + tuple(
                (dict(kv.split("=") for kv in e.split(",")), 16) for e in filter(None, os.environ.get("EXTRA", "").split(";"))):
            time.sleep(0.5)
            env = dict(os.environ, DSH_TIMING="1", DSH_T0=repr(time.time()), **env_extra)
            t0 = time.perf_counter()
            r = subprocess.run([cli, "dist", "-k", "31", "-S", "10", "-p", str(threads), "-b", "--avoid-sorting", "-O", out, "-o", os.devnull, "-F", lst],
                               capture_output=True, env=env, timeout=600)
            wall = time.perf_counter() - t0
            data = open(out, "rb").read() if r.returncode == 0 else None
            if ref is None:
                ref = data
            print(json.dumps({"env": env_extra, "threads": threads, "rc": r.returncode, "wall_s": round(wall, 4), "same_matrix_as_first_run": data == ref,
                              "timing": [l for l in r.stderr.decode(errors="replace").splitlines() if "[timing]" in l][:8] + [l for l in r.stderr.decode(errors="replace").splitlines() if "[timing]" in l][-3:]}), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
