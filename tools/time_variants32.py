import sys, shutil, subprocess
for lib in sys.argv[1:]:
    shutil.copy(lib, "sqfa_amd/lib/libsqfa_hip.so")
    out = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.'); import torch, time_pairs as t; t.run(1000,32,False,torch.float32,reps=3)"], capture_output=True, text=True)
    print(lib, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-500:], flush=True)
