"""Drop-in `pointops2_cuda` extension module for a DODA checkout."""
from doda_amd.pointops2_cuda import *  # noqa: F401,F403
from doda_amd.pointops2_cuda import knnquery_cuda  # noqa: F401
