"""Recipe for oracle/_ref/ref_glue: the reference's own StereoConvolver.cpp compiled WHERE IT LIES against the drop-in
Convolver (include/reevr_amd/Convolver.h, C ABI underneath) plus the test driver tests/ref_glue_main.cpp. Test
infrastructure like everything under oracle/: only tests/ and __graft_entry__.build() call it; the product never does.
Needs /root/reference (this container); the GPU box uses the prebuilt binary."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SC = "/root/reference/src/dsp/StereoConvolver.cpp"


GLUE_EXE = os.path.join(ROOT, "oracle", "_ref", "ref_glue")     # (git-ignored, travels to the GPU box like oracle/_ref's other files)


def build_reference_glue(tmp, exe=None):
    """"the reference's StereoConvolver.cpp then compiles unchanged against include/reevr_amd/Convolver.h".
    The reference's StereoConvolver.{h,cpp} are compiled WHERE THEY LIE -- reached through symlinks in a temp dir, because a
    quoted #include looks in the including file's own directory first and would find the JUCE-bound src/dsp/Convolver.h --
    next to three temp-dir headers: Convolver.h = the drop-in, JuceHeader.h = the std headers the file relies on, Impulse.h =
    the four buffers + isQuad loadImpulse reads and the SVF::EQBand the header names. Nothing of the reference is copied."""
    import subprocess
    from reevr_amd import _lib, build
    build.build_lib()
    os.symlink(REF_SC, os.path.join(tmp, "StereoConvolver.cpp"))
    os.symlink(REF_SC[:-3] + "h", os.path.join(tmp, "StereoConvolver.h"))
    open(os.path.join(tmp, "JuceHeader.h"), "w").write("#pragma once\n#include <algorithm>\n#include <memory>\n#include <vector>\n")
    open(os.path.join(tmp, "Convolver.h"), "w").write('#pragma once\n#include "reevr_amd/Convolver.h"\n')
    open(os.path.join(tmp, "Impulse.h"), "w").write(
        "#pragma once\n#include <vector>\n"
        "struct SVF { enum Mode { LP, BP, HP, LS, HS, PK, BS, HP6, LP6, Off }; struct EQBand { Mode mode; float freq, q, gain; }; };\n"
        "struct Impulse { std::vector<float> bufferLL, bufferRR, bufferLR, bufferRL; bool isQuad = false; };\n")
    exe = exe or os.path.join(tmp, "ref_glue")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", tmp, "-I", os.path.join(ROOT, "include"),
                    os.path.join(tmp, "StereoConvolver.cpp"), os.path.join(ROOT, "tests", "ref_glue_main.cpp"), "-o", exe,
                    "-L", libdir, "-lreevr_amd", f"-Wl,-rpath,{libdir}"], check=True)
    return exe
