"""Build ablation / experiment variants of libreevr_amd.so into abl_libs/<name>/ (git-ignored, travels
to the GPU box):   python tools/abl_build.py name:-DFLAG1,-DFLAG2 name2:-DX ...
Run a variant with  REEVR_AMD_LIB=abl_libs/<name>/libreevr_amd.so python tools/kern_times.py 2"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reevr_amd import build  # noqa: E402


def one(spec):
    name, _, flags = spec.partition(":")
    out = os.path.join(ROOT, "abl_libs", name)
    os.makedirs(out, exist_ok=True)
    objs = build.compile_objects(os.path.join(out, "obj"), defines=["-DRVC_DEV_BUILD", "-w"] + [f for f in flags.split(",") if f])
    build.link_lib(objs, os.path.join(out, "libreevr_amd.so"))
    return name


if __name__ == "__main__":
    with ThreadPoolExecutor(8) as ex:
        for n in ex.map(one, sys.argv[1:]):
            print("built", n)
