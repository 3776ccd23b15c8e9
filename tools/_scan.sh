for rep in 1 2 3; do for f in 8 0; do echo -n "flags=$f: "; python - <<PY
import sys,subprocess,json,os
sys.path.insert(0,'tools'); sys.path.insert(0,'.')
from clusterfusion_amd import _lib
_lib.load().cf_debug_set_flags($f)
import config_bench
PY
done; done
