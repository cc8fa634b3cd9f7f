"""Diagnostic: python tools/bench_with_norm_mode.py <mode> <bench args...>  (ChannelNorm backward:
0 = gy in registers + xhat in LDS (one workgroup per CU); 1 = stream-twice (default))"""
import os
# needs the development build of the library (make -C paradis_model_amd/csrc dev): the shipped one exports
# no paradis_debug_set_* tunables
os.environ.setdefault("PARADIS_HIP_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                                      "paradis_model_amd", "libparadis_hip_dev.so"))

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paradis_model_amd._lib import lib
lib.paradis_debug_set_norm_bwd_reread(int(sys.argv[1]))
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
