set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python - <<'PY'
import os, sys, pathlib, tempfile
ROOT=os.environ["GRAFT_REPO_ROOT"]; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
import test_mini_stereo_gpu as t
tmp = pathlib.Path("/tmp/stereo_prof_ds"); 
import shutil; shutil.rmtree(tmp, ignore_errors=True); tmp.mkdir()
cmd, out_dir, _, _ = t.make_stereo_dataset(tmp, 30, seed=171, ds="dsA")
open("/tmp/stereo_cmd.txt","w").write("\n".join(cmd))
PY
mapfile -t CMD < /tmp/stereo_cmd.txt
rm -rf /tmp/prof_st && rocprofv3 --hip-trace --stats --output-format csv -d /tmp/prof_st -- "${CMD[@]}" 30 8 0.5 1 1 1 > /tmp/prof_st.log 2>&1 || { tail -20 /tmp/prof_st.log; exit 1; }
f=$(find /tmp/prof_st -name "*hip_api_stats.csv" | head -1)
head -25 $f > gpurun_out/r06/stereo_ls_S1_hip_api_stats.csv
t=$(find /tmp/prof_st -name "*hip_api_trace.csv" | head -1); cp $t gpurun_out/r06/stereo_ls_S1_hip_api_trace.csv; wc -l gpurun_out/r06/stereo_ls_S1_hip_api_trace.csv
