"""The CPU oracle in float64 -- TEST INFRASTRUCTURE ONLY (see oracle.py).  Same source, executed under this module's name: ``REAL`` = float64 and
``libvt_oracle64.so`` (vt_oracle.c built with -DVTO_FP64).  Used as the arbiter of long Adam trajectories: a HIP result is accepted when it is as close
to this run as the fp32 oracle is (or within the 1e-3 m bar)."""
import os as _os

with open(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "oracle.py")) as _f:
    exec(compile(_f.read(), _f.name, "exec"), globals())
