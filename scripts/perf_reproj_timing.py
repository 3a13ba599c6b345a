import sys, os, subprocess, pathlib, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_mini_frontend_gpu as t
tmp = pathlib.Path(tempfile.mkdtemp(prefix="chain_", dir="/tmp"))
cmd, out_dir, poses, stamps, n_frames = t.make_dataset(tmp)
for k in range(2):
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, SVOH_REPROJ_TIMING="1"))
print(r.stdout[-600:]); print(r.stderr[-1500:])
