"""Time the fused loss+grad launch for several pre-built library variants in one GPU session:
    python tools/time_variants_any.py M DTYPE lib1.so lib2.so ...      (DTYPE: f32 | f64)
Each variant runs in its own process (SQFA_HIP_LIBRARY selects it; the installed library is not touched)."""
import os, subprocess, sys
m, dt = int(sys.argv[1]), sys.argv[2]
reps = 8 if m <= 17 else 3
for lib in sys.argv[3:]:
    code = ("import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.'); import torch, time_pairs as t; "
            f"t.run(1000,{m},False,torch.float{'32' if dt == 'f32' else '64'},reps={reps})")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                         env=dict(os.environ, SQFA_HIP_LIBRARY=os.path.abspath(lib)))
    line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-500:]
    print(lib, line[:110], flush=True)
