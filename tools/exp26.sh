# does the sleeping wait survive a process where torch has initialised the device first?  (bench.py with N > 1 does)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python - <<'PY' 2>&1 | grep "A: width\|call \|torch" | cut -c1-120
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
print("torch initialised the device first:", torch.cuda.is_initialized())
sys.argv = ["gap_paths.py", "4194304", "trace=1"]
exec(open("tools/gap_paths.py").read())
PY
