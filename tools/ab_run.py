# python tools/ab_run.py <lib.so> <tool.py> [args...]: run a tools/ script against another build of the library
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ctypes as C
from brickmap_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
probe = C.CDLL(_lib.LIB_PATH)
for name in list(_lib.SIGNATURES):
    if not hasattr(probe, name):
        del _lib.SIGNATURES[name]
tool = os.path.join(ROOT, sys.argv[2])
sys.argv = [tool] + sys.argv[3:]
__file__ = tool
exec(open(tool).read())
