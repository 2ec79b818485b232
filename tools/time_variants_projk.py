"""Run tools/time_projection_kernel.py for several pre-built library variants (SQFA_HIP_LIBRARY selects each; the installed
library is not touched)."""
import os, subprocess, sys
for lib in sys.argv[1:]:
    out = subprocess.run([sys.executable, "tools/time_projection_kernel.py"], capture_output=True, text=True,
                         env=dict(os.environ, SQFA_HIP_LIBRARY=os.path.abspath(lib)))
    print(lib); print("\n".join(l for l in out.stdout.splitlines() if l.startswith("C=")), flush=True)
