"""Lock-step front end (host/svo_hip_lockstep.h) on the synthetic EuRoC-layout sequence of tests/test_mini_frontend_gpu.py:
frames/s for several (streams, host threads, groups), next to the thread-per-stream mode of round 4.
  python scripts/perf_lockstep.py [configs "S:W:G,S:W:G,..."] [n_laps]
Prints the tool's own line per configuration (SVOH_LOCKSTEP_TIMING=1: the phases of a round)."""
import os
import pathlib
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_mini_frontend_gpu as t
    configs = sys.argv[1] if len(sys.argv) > 1 else "1:1:1,8:4:1,32:8:1,32:16:1,32:8:2,64:8:2"
    n_laps = sys.argv[2] if len(sys.argv) > 2 else "3"
    with tempfile.TemporaryDirectory() as d:
        cmd, out_dir, poses, stamps, n_frames = t.make_dataset(pathlib.Path(d))
        env = dict(os.environ)
        env["SVOH_LOCKSTEP_TIMING"] = "1"
        if os.environ.get("PERF_THREADS", "1") != "0":
            for n in (1, 8):
                r = subprocess.run(cmd + [str(n_frames), "8", str(n)], capture_output=True, text=True)
                print("threads mode, %d stream(s): %s" % (n, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr), flush=True)
        for c in configs.split(","):
            s, w, g = c.split(":")
            r = subprocess.run(cmd + [str(n_frames), "8", s, "lockstep", w, g, n_laps], capture_output=True, text=True, env=env)
            print(r.stdout.strip(), flush=True)
            print(r.stderr.strip(), flush=True)
            if r.returncode != 0:
                print("FAILED rc=%d" % r.returncode)
                continue
            import numpy as np
            fc = np.loadtxt(str(out_dir / "frontend.csv"), delimiter=",", skiprows=1)[3:]
            names = "pyramid align reproject pose seeds keyframe total".split()
            # a keyframe round pays the detector; the round AFTER it aligns against a keyframe (cluster geometry, 3x the patches)
            kf = fc[:, 1] > 0
            after = np.roll(kf, 1); after[0] = False
            for label, sel in (("plain rounds", ~kf & ~after), ("keyframe rounds", kf), ("rounds after a keyframe", after & ~kf)):
                if sel.any():
                    print("   %-24s median ms: " % label + " ".join("%s %.3f" % (n, v) for n, v in zip(names, np.median(fc[sel][:, 7:14], axis=0))), flush=True)


if __name__ == "__main__":
    main()
