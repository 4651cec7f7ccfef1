"""TEST INFRASTRUCTURE: compile kernel sources of beyond_deep_ensembles_amd/csrc for the CPU execution model of
hip_emu.hpp.  The sources are used as they are, except for what a host compiler cannot parse:

  * `#include <hip/hip_runtime.h>`              -> `#include "hip_emu.hpp"`
  * `extern __shared__ [aligned] T name[];`     -> `T* name = static_cast<T*>(hip_emu::dyn_lds());`
  * `asm volatile("s_waitcnt vmcnt(N)" ...);`   -> `hip_emu::waitcnt_vm(N); hip_emu::wave_sync();`  (this lane's LDS-DMA
                                                   requests land here, not before; then the wave meets)
  * `asm volatile("s_waitcnt lgkmcnt(0)" ...);` -> `hip_emu::wave_sync();`  -- the lanes of a wave are fibers that run from
                                                   rendezvous to rendezvous, not in lockstep, so a hand-off through LDS
                                                   between lanes of ONE wave needs a meeting point; the kernels written
                                                   for this (conv_lrt*.hip, swag_batched.hip) carry an explicit s_waitcnt
                                                   at exactly those places (the implicit hand-offs of the other kernels
                                                   are covered by the lockstep-at-LDS-access scheduling, hip_emu.cpp)
  * every other `asm volatile(...)` statement   -> dropped (register-class barriers)

The result is cached under tests/hip_emu/_build/ keyed by the content of every input."""
import contextlib
import hashlib
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "beyond_deep_ensembles_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
HEADERS = ["bde_common.hpp", "conv_common.hpp", "svgd_shared.hpp", "svgd_gram.hpp"]

_EXTERN_LDS = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?(\w+)\s+(\w+)\[\];")
_ASM = re.compile(r"asm\s+volatile\s*\((?:[^()]|\((?:[^()]|\([^()]*\))*\))*\)\s*;")


def _asm(match):
    text = match.group(0)
    m = re.search(r"vmcnt\((\d+)\)", text)
    if m:
        return f"hip_emu::waitcnt_vm({m.group(1)}); hip_emu::wave_sync();"
    if "lgkmcnt" in text:
        return "hip_emu::wave_sync();"
    return ";"


def transform(text, name):
    text = text.replace("#include <hip/hip_runtime.h>", '#include "hip_emu.hpp"')
    text = text.replace('#include "../../include/bde_hip.h"', f'#include "{os.path.join(ROOT, "include", "bde_hip.h")}"')
    text = _EXTERN_LDS.sub(lambda m: f"{m.group(1)}* {m.group(2)} = static_cast<{m.group(1)}*>(hip_emu::dyn_lds());", text)
    text = _ASM.sub(_asm, text)
    return f'#line 1 "{os.path.join(CSRC, name)}"\n' + text


def _prune(root, keep):
    """Builds of older source states: keep the most recently used few (a full build is ~90 MB of objects)."""
    import shutil
    try:
        dirs = sorted((os.path.join(root, d) for d in os.listdir(root) if not d.startswith(".")), key=os.path.getmtime, reverse=True)
    except OSError:
        return
    for d in dirs[keep:]:
        shutil.rmtree(d, ignore_errors=True)


def available():
    return os.path.exists(CLANG) and os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h")


def build(sources, defines=()):
    """-> path of a shared library holding the C-ABI entry points of `sources` (names under csrc/) on the CPU model."""
    names = list(sources) + HEADERS
    texts = {n: transform(open(os.path.join(CSRC, n)).read(), n) for n in names}
    emu = [open(os.path.join(HERE, f)).read() for f in ("hip_emu.hpp", "hip_emu.cpp")]
    key = hashlib.sha256(("\0".join(texts[n] for n in names) + "\0".join(emu) + repr(tuple(defines)) + open(__file__).read()).encode()).hexdigest()[:16]
    out_dir = os.path.join(HERE, "_build", key)
    lib = os.path.join(out_dir, "libbde_emu.so")
    if os.path.exists(lib):
        return lib
    with _locked():                                   # several test processes (two-rank runs, xdist) may want it at once
        return _build_locked(sources, defines, names, texts, out_dir, lib)


@contextlib.contextmanager
def _locked():
    import fcntl
    os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
    with open(os.path.join(HERE, "_build", ".lock"), "w") as fh:
        fcntl.flock(fh, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(fh, fcntl.LOCK_UN)


def _build_locked(sources, defines, names, texts, out_dir, lib):
    if os.path.exists(lib):
        return lib
    _prune(os.path.join(HERE, "_build"), keep=4)
    os.makedirs(out_dir, exist_ok=True)
    for n in names:
        with open(os.path.join(out_dir, n.replace(".hip", ".cpp") if n.endswith(".hip") else n), "w") as fh:
            fh.write(texts[n])
    flags = ["-x", "c++", "-std=c++17", "-O1", "-g", "-fPIC", "-ffp-contract=off", "-Wno-unused-value", "-Wno-unknown-attributes",
             "-Wno-ignored-attributes", "-I", HERE, "-I", "/opt/rocm/include", "-I", out_dir] + [f"-D{d}" for d in defines]
    # the kernel sources (not hip_emu.cpp) get clang's thread-sanitizer INSTRUMENTATION only; hip_emu.cpp implements the
    # callbacks (lockstep of a wave's lanes at LDS accesses), the sanitizer runtime is not linked
    tsan = ["-fsanitize=thread", "-mllvm", "-tsan-instrument-func-entry-exit=0", "-mllvm", "-tsan-instrument-atomics=0",
            "-mllvm", "-tsan-instrument-memintrinsics=0", "-mllvm", "-tsan-instrument-read-before-write=1"]
    objs = []
    procs = []
    for n in list(sources) + ["hip_emu.cpp"]:
        src = os.path.join(HERE, n) if n == "hip_emu.cpp" else os.path.join(out_dir, n.replace(".hip", ".cpp"))
        obj = os.path.join(out_dir, os.path.basename(src) + ".o")
        objs.append(obj)
        extra = [] if n == "hip_emu.cpp" else tsan
        procs.append((n, subprocess.Popen([CLANG] + flags + extra + ["-c", src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for n, p in procs:
        log = p.communicate()[0].decode()
        if p.returncode:
            raise RuntimeError(f"hip_emu build of {n} failed:\n{log[-6000:]}")
    tmp = lib + ".tmp"
    # -Bsymbolic: the process may already hold libbde_hip.so / libamdhip64.so (RTLD_GLOBAL) with the same symbol names;
    # every reference inside this library must bind to its own definitions
    subprocess.check_call([CLANG, "-shared", "-Wl,-Bsymbolic", "-o", tmp] + objs + ["-lpthread"])
    os.replace(tmp, lib)
    return lib


def build_host_nodes(sources):
    """-> path of `_bde_host_emu.so`: the C++ autograd nodes of csrc/host_autograd.cpp (what lib/_bde_host.so binds for the
    Bayesian layers) over the CPU model.  The source is used as it is except for its two device-specific helpers: the
    current-stream getter returns NULL and the "is a CUDA tensor" check asks for a CPU tensor.  The kernel objects of
    build(sources) are linked INTO the module (-Bsymbolic), so its C-ABI calls cannot bind to a libbde_hip.so the process
    may also hold."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    lib = build(sources)
    out_dir = os.path.dirname(lib)
    text = open(os.path.join(CSRC, "host_autograd.cpp")).read()
    old_stream = "void* current_stream(const at::Tensor& t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }"
    assert old_stream in text and "t.is_cuda() &&" in text and "#include <c10/hip/HIPStream.h>\n" in text
    text = text.replace(old_stream, "void* current_stream(const at::Tensor&) { return nullptr; }")
    text = text.replace("t.is_cuda() &&", "!t.is_cuda() &&").replace("#include <c10/hip/HIPStream.h>\n", "")
    text = text.replace('#include "../../include/bde_hip.h"', f'#include "{os.path.join(ROOT, "include", "bde_hip.h")}"')
    text += "\nPYBIND11_MODULE(_bde_host_emu, m) { bind_autograd_nodes(m); }\n"
    key = hashlib.sha256((text + torch.__version__).encode()).hexdigest()[:16]
    out = os.path.join(out_dir, f"_bde_host_emu_{key}.so")
    if os.path.exists(out):
        return out
    with _locked():
        if os.path.exists(out):
            return out
        return _build_host_nodes_locked(sources, text, out_dir, out)


def _build_host_nodes_locked(sources, text, out_dir, out):
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    src = os.path.join(out_dir, "host_autograd_emu.cpp")
    with open(src, "w") as fh:
        fh.write(text)
    objs = [os.path.join(out_dir, n.replace(".hip", ".cpp") + ".o") for n in sources] + [os.path.join(out_dir, "hip_emu.cpp.o")]
    libs = ce.library_paths()
    cmd = ["g++", "-O1", "-std=c++17", "-shared", "-fPIC", src] + objs + ["-o", out + ".tmp", "-I" + sysconfig.get_paths()["include"],
           "-DTORCH_EXTENSION_NAME=_bde_host_emu", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-Wl,-Bsymbolic"]
    cmd += ["-I" + i for i in ce.include_paths()] + ["-L" + l for l in libs]
    cmd += ["-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10", "-lpthread", "-Wl,-rpath," + libs[0]]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if p.returncode:
        raise RuntimeError("hip_emu build of the host autograd nodes failed:\n" + p.stdout.decode()[-6000:])
    os.replace(out + ".tmp", out)
    return out


def load_host_nodes(sources):
    import importlib.util
    import torch  # noqa: F401
    path = build_host_nodes(sources)
    spec = importlib.util.spec_from_file_location("_bde_host_emu", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import ctypes
    LOADED[path] = ctypes.CDLL(path)          # the module carries its own copy of the model: its launches count too
    return mod


LOADED = {}            # path -> ctypes handle of every CPU-model library this process has loaded (emu_ops.launched_kernels)


if __name__ == "__main__":
    import sys
    print(build(sys.argv[1:] or ["conv_lrt.hip", "conv_lrt_bwd.hip"]))
