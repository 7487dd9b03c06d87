"""Properties of the built device code that the hand-scheduled kernels rely on (CPU only: reads the objects the build made).

The pipe kernels (kernels_arb_pipe.hip, kernels_farrow_pipe.hip) issue their LDS reads from inline assembly and wait for them
with counted s_waitcnt: between issue and wait the destination registers are NOT valid, which the compiler does not know.  A
register spill inside that window would save a register before its data has landed (round 3 met the same hazard with an
asynchronous s_load).  So no instantiation of these kernels may use scratch memory at all."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multirate.jl_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"


def _kernel_scratch(obj):
    """{kernel name: private_segment_fixed_size + agpr_count} of the gfx950 code object bundled in a host object file (the bundle is
    compressed: --offload-compress; clang-offload-bundler unpacks both kinds)."""
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "co.o")
        got = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(tmp, "copy.o")], capture_output=True, text=True, timeout=300)
        assert got.returncode == 0 and os.path.exists(fat), "no device code bundled in " + obj      # (a host-only unit: check_spilling_instantiations.py skips it)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co],
                       capture_output=True, text=True, timeout=300, check=True)
        assert os.path.getsize(co) > 0, "no device code object bundled in " + obj
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, timeout=300, check=True).stdout
    out = {}
    for entry in notes.split("\n  - .agpr_count")[1:]:
        name = re.search(r"\n    \.name:\s+(\S+)", entry)
        size = re.search(r"\n    \.private_segment_fixed_size:\s+(\d+)", entry)
        agpr = re.match(r":\s+(\d+)", entry)                     # (AccVGPRs: the compiler's other place to park registers)
        if name and size and agpr:
            out[name.group(1)] = int(size.group(1)) + int(agpr.group(1))
    return out


@pytest.mark.parametrize("src,kernel", [("kernels_arb_pipe.hip", "arb_pipe_kernel"), ("kernels_farrow_pipe.hip", "farrow_pipe_kernel")])
def test_pipe_kernels_use_no_scratch(pkg, src, kernel):
    obj = os.path.join(CSRC, "build", src + ".o")
    if not os.path.exists(obj) or not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("no built object (the library came prebuilt) or no llvm tools")
    if os.path.getmtime(obj) < os.path.getmtime(os.path.join(CSRC, src)):
        pytest.skip("object older than its source")
    mk = open(os.path.join(CSRC, "Makefile")).read()
    # the flag that keeps the Float32 instantiations within their register budget
    assert re.search(r"kernels_arb_pipe\.hip\.o[^\n]*kernels_farrow_pipe\.hip\.o[^\n]*-fno-slp-vectorize", mk), \
        "Makefile no longer builds the pipe kernels with -fno-slp-vectorize"
    sizes = {k: v for k, v in _kernel_scratch(obj).items() if kernel in k}
    assert len(sizes) >= 12, f"expected the instantiations of {kernel} in the object, found {len(sizes)}"
    spilling = {k: v for k, v in sizes.items() if v != 0}
    assert not spilling, f"{kernel}: instantiations with scratch or AccVGPRs (registers parked next to hand-issued LDS reads): {list(spilling.items())[:6]}"


def test_library_ships_compressed_device_code(pkg):
    """The ~700 kernel instantiations are 46 MB of gfx950 code uncompressed; every translation unit is built with --offload-compress
    (the HIP runtime unpacks a unit's bundle when its first kernel is launched): the library stays under 20 MB."""
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert re.search(r"^CXXFLAGS\s*=.*--offload-compress", mk, re.M), "Makefile no longer compresses the device code"
    lib = os.path.join(os.path.dirname(CSRC), "libmultirate_hip.so")
    assert os.path.exists(lib)
    assert os.path.getsize(lib) < 20_000_000, os.path.getsize(lib)


def test_lane_kernel_uses_no_scratch_and_its_statements_are_the_generators(pkg):
    """arb_lane_kernel (kernels_arb_lane.hip): its pair statements keep tap blocks in fixed scalar registers and samples in registers
    whose reads are in flight inside the statement only; staging keeps global loads in flight across steps -- a scratch reload waits
    for every one of them (s_waitcnt vmcnt(0)): no instantiation may use scratch.  The statements (arb_lane_pair.inc) are generated:
    the committed file must be what scripts/gen_arb_lane_asm.py writes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_arb_lane_asm", os.path.join(ROOT, "scripts", "gen_arb_lane_asm.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert open(os.path.join(CSRC, "arb_lane_pair.inc")).read() == gen.render(), "arb_lane_pair.inc is stale: run scripts/gen_arb_lane_asm.py"
    obj = os.path.join(CSRC, "build", "kernels_arb_lane.hip.o")
    if not os.path.exists(obj) or not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("no built object (the library came prebuilt) or no llvm tools")
    if os.path.getmtime(obj) < max(os.path.getmtime(os.path.join(CSRC, f)) for f in ("kernels_arb_lane.hip", "arb_lane_pair.inc")):
        pytest.skip("object older than its source")
    sizes = {k: v for k, v in _kernel_scratch(obj).items() if "arb_lane_kernel" in k}
    assert len(sizes) == 4, f"expected STRICT / FUSED x 32 / 16 taps per phase, found {sorted(sizes)}"
    assert not {k: v for k, v in sizes.items() if v != 0}, sizes


def test_interp_lane_kernel_uses_no_scratch_and_its_statements_are_the_generators(pkg):
    """interp_lane_kernel (kernels_interp_lane.hip) keeps its channels' sliding window in 78 VGPRs next to the unit of samples in flight:
    at three waves per SIMD (168 VGPRs) nothing may spill (at four, 128, it does: 3.9 ms against 2.85 on config 3a).  Its statements
    (interp_lane_quad.inc) are generated: the committed file must be what scripts/gen_interp_lane_asm.py writes; a phase's sum and
    product sit on different VGPR banks."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_interp_lane_asm", os.path.join(ROOT, "scripts", "gen_interp_lane_asm.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert open(os.path.join(CSRC, "interp_lane_quad.inc")).read() == gen.render(), "interp_lane_quad.inc is stale: run scripts/gen_interp_lane_asm.py"
    spec = importlib.util.spec_from_file_location("gen_arb_window_asm", os.path.join(ROOT, "scripts", "gen_arb_window_asm.py"))
    genw = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(genw)
    assert open(os.path.join(CSRC, "arb_window_one.inc")).read() == genw.render(), "arb_window_one.inc is stale: run scripts/gen_arb_window_asm.py"
    spec = importlib.util.spec_from_file_location("gen_decim_lane_asm", os.path.join(ROOT, "scripts", "gen_decim_lane_asm.py"))
    gend = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gend)
    assert open(os.path.join(CSRC, "decim_lane_group.inc")).read() == gend.render(), "decim_lane_group.inc is stale: run scripts/gen_decim_lane_asm.py"
    for p in range(4):
        acc, tmp = gen.acc_reg(4, p)
        assert (acc % 4 < 2) != (tmp % 4 < 2), (p, acc, tmp)
    obj = os.path.join(CSRC, "build", "kernels_interp_lane.hip.o")
    if not os.path.exists(obj) or not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("no built object (the library came prebuilt) or no llvm tools")
    if os.path.getmtime(obj) < max(os.path.getmtime(os.path.join(CSRC, f)) for f in ("kernels_interp_lane.hip", "interp_lane_quad.inc")):
        pytest.skip("object older than its source")
    sizes = {k: v for k, v in _kernel_scratch(obj).items() if "interp_lane_kernel" in k}
    assert len(sizes) == 2, f"expected STRICT and FUSED, found {sorted(sizes)}"
    assert not {k: v for k, v in sizes.items() if v != 0}, sizes


def test_stream_pair_statements_are_the_generators(pkg):
    """fir_stream_kernel's hand-scheduled pair for config 3b's shape (fir_stream_pair_c64_m4.inc) is generated: the committed file must be
    what scripts/gen_fir_stream_asm.py writes; its sample buffers and sums live in fixed VGPRs below 72 (the kernel keeps 7 waves per SIMD)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_fir_stream_asm", os.path.join(ROOT, "scripts", "gen_fir_stream_asm.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert open(os.path.join(CSRC, "fir_stream_pair_c64_m4.inc")).read() == gen.render(), "fir_stream_pair_c64_m4.inc is stale: run scripts/gen_fir_stream_asm.py"
    assert max(gen.V_CLOB) < 72 and len(set(a % 4 for pair in gen.ACC.values() for a in pair)) == 2     # sums and products on different banks


@pytest.mark.gpu
def test_instantiations_with_scratch_are_bit_exact_on_the_gpu():
    """The older hand-scheduled kernels (output-pair, streaming) have instantiations that do use scratch memory: each one, read
    from the built objects, runs against the universal kernel bit for bit (scripts/check_spilling_instantiations.py)."""
    import glob
    import sys
    if not glob.glob(os.path.join(CSRC, "build", "kernels_*.hip.o")) or not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("no built objects on this box (the library came prebuilt) or no llvm tools")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_spilling_instantiations.py")], capture_output=True, text=True, timeout=900)
    tail = "\n".join(p.stdout.splitlines()[-6:])
    assert p.returncode == 0, tail + p.stderr[-2000:]
    m = re.search(r"instantiations with scratch: (\d+) mismatches: (\d+)", p.stdout)
    assert m and int(m.group(2)) == 0, tail
