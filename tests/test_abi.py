"""CPU: libmla_hip.so builds for gfx950, loads, and exports every symbol include/mla_hip.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "mla_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mla_[a-z0-9_]+)\s*\(", txt)) - {"mla_stream_t"})


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from mla_amd import hip
    lib = ctypes.CDLL(hip._LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 45
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in include/mla_hip.h but not exported: {missing}"
    # the ctypes binding and the header agree on the symbol set
    assert set(hip.exported_symbols()) == set(syms), set(hip.exported_symbols()) ^ set(syms)


def test_attention_object_allocates_nothing_and_keeps_no_state():
    """kernels never allocate, no library-owned memory, no static scratch. Round 5's merged backward launch broke that in
    one entry point (hipMalloc'ed per-stream counters behind a static std::map + mutex); round 6 made the counters caller-owned. Proved
    on the object file: attention.o references no allocation / memset / memcpy entry point of the HIP runtime and no std::map / mutex."""
    import subprocess
    import __graft_entry__ as g
    g.build()
    obj = os.path.join(ROOT, "mla_amd", "csrc", "build", "attention.o")
    und = subprocess.run(["nm", "-u", "-C", obj], check=True, capture_output=True, text=True).stdout
    bad = [ln.strip() for ln in und.splitlines()
           if re.search(r"hipMalloc|hipFree|hipMemset|hipMemcpy|hipHostMalloc|std::_Rb_tree|pthread_mutex|std::mutex", ln)]
    assert not bad, bad
    # the product library does not carry the attention experiment kernels (attention_exp.inc is compiled with MLA_EXPERIMENTAL=1 only)
    so = open(os.path.join(ROOT, "mla_amd", "libmla_hip.so"), "rb").read()
    for name in (b"attn_bwd_fused_kernel", b"attn_bwd_dq5_kernel", b"attn_delta_kernel"):
        assert name not in so, name
    # argument validation of the caller-owned counters happens on the host
    from mla_amd import hip
    lib = hip.lib()
    assert lib.mla_attn_bwd_sync_ints(32, 32) == 2048 and lib.mla_attn_bwd_sync_ints(1, 3) == 16 and lib.mla_attn_bwd_sync_ints(0, 3) == -1
    P = ctypes.c_void_p(16)
    rc = lib.mla_attn_bwd(P, P, P, P, P, P, None, P, P, P, P, 1, 64, 3, 128, 1152, 384, 0.1, None, None, ctypes.c_void_p(64), 8, None)
    assert rc < 0 and b"head_sync needs" in lib.mla_last_error(), lib.mla_last_error()


def test_query_and_error_plumbing():
    from mla_amd import hip
    lib = hip.lib()
    assert lib.mla_query(0) == 1 and lib.mla_query(1) == 950 and lib.mla_query(2) == 64 and lib.mla_query(99) == -1
    # argument validation happens on the host before any launch -> safe without a GPU
    rc = lib.mla_gemm_bf16(None, None, None, None, None, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 1.0, 0, None)
    assert rc < 0 and b"null operand" in lib.mla_last_error()
    rc = lib.mla_attn_fwd(ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), None,
                          1, 8, 1, 64, 192, 64, 0.125, None)
    assert rc < 0 and b"head_dim must be 128" in lib.mla_last_error()


def test_no_cpu_fallback():
    """The product path fails loudly without a GPU instead of silently computing on the CPU."""
    import pytest
    import torch
    from mla_amd import ops
    with pytest.raises((RuntimeError, TypeError)):
        ops.linear(torch.zeros(4, 8, dtype=torch.bfloat16), (torch.zeros(8, 8, dtype=torch.bfloat16),))
    with pytest.raises((RuntimeError, TypeError)):
        ops.rmsnorm(torch.zeros(4, 8), torch.ones(8), 1e-5)


def test_generated_assembly_is_what_the_generator_writes(tmp_path):
    """The committed main-loop assembly (gemm256_kloop*.inc = the product GEMM loop, gemm_asm_*.inc = the experiment kernels) is exactly what
    tools/gen_gemm_asm.py emits with its default flags: nobody edits the .inc files by hand, nobody changes the generator without
    regenerating them."""
    import subprocess
    import sys
    env = dict(os.environ, GEN_OUT_DIR=str(tmp_path))
    for k in ("GEN256_FLAGS", "GEN8W_FLAGS", "GEN4W_FLAGS"):
        env.pop(k, None)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_gemm_asm.py")], check=True, env=env, capture_output=True)
    names = ["gemm256_kloop.inc", "gemm256_kloop_half1.inc", "gemm256_kloop_clobbers.inc", "gemm_asm_8w_loop.inc", "gemm_asm_4w_loop.inc",
             "gemm_asm_8w_clobbers.inc", "gemm_asm_4w_clobbers.inc"]
    for n in names:
        new, old = open(tmp_path / n).read(), open(os.path.join(ROOT, "mla_amd", "csrc", n)).read()
        assert new == old, f"{n} differs from the generator's output: run python tools/gen_gemm_asm.py"
    # the product loop's shape: three barriers per K-tile in the loop body, no vmcnt(0) inside it, at most two non-MFMA instructions per gap
    body = [ln.strip('"\\n \n') for ln in open(tmp_path / "gemm256_kloop.inc") if ln.startswith('"')]
    lo, hi = body.index("1:"), body.index("4:")
    loop = body[lo + 1:hi]
    assert sum(ln == "s_barrier" for ln in loop) == 6 and not any(ln.startswith("s_waitcnt vmcnt(0)") for ln in loop)      # two K-tiles per iteration
    assert sum(ln.startswith("v_mfma") for ln in loop) == 128 and sum(ln.startswith("global_load_lds") for ln in loop) == 16
    gap = worst = 0
    for ln in loop:
        if ln.startswith("v_mfma"):
            gap = 0
        elif not ln.startswith(";") and not ln.startswith("s_sub_u32 s46") and not ln.startswith("s_cmp") and not ln.startswith("s_cbranch"):
            gap += 1
            worst = max(worst, gap)
    assert worst <= 2, worst


def test_kloop_checker_flags_an_accumulator_touch_between_two_statements(tmp_path):
    """tools/check_kloop_asm.py (run by build.sh on the real device assembly): a compiler instruction that touches an accumulation
    register between two inline-asm statements of an assembly-loop kernel is a finding; after a single statement it is not; a spill
    always is."""
    import subprocess
    import sys
    mfma = "v_mfma_f32_16x16x32_bf16 a[0:3], v[0:3], v[4:7], a[0:3]\n"
    def head(epi, n_mfma=128, indent=""):
        return (f"_ZN1x14gemm256_kernelILi0ELi0ELi{epi}ELb1EEEv8GemmArgs:\n\ts_load_dword s0, s[0:1], 0x0\n" + (indent + mfma) * n_mfma)
    tail = "\ts_endpgm\n.Lfunc_end0:\n"
    cases = {"two_statements": (head(0) + "\tv_accvgpr_write_b32 a5, v1\nv_accvgpr_read_b32 v4, a64\n" + tail, 1),
             "two_statements_clean": (head(0) + "\tv_mov_b32 v5, v1\nv_accvgpr_read_b32 v4, a64\n" + tail, 0),
             "one_statement": (head(1) + "\tv_accvgpr_write_b32 a5, v1\n" + tail, 0),
             "spill": (head(1) + "\tscratch_store_dword off, v1, s0\n" + tail, 1),
             # advisor (round 3): the checks must not pass vacuously when the listing format hides the inline assembly
             "plain_kernel_with_one_statement": (head(0) + "\tv_mov_b32 v5, v1\n" + tail, 1),
             "indented_inline_asm": (head(1, indent="\t") + tail, 1),
             "too_few_mfma_in_asm": (head(1, n_mfma=8) + tail, 1)}
    for name, (src, want) in cases.items():
        f = tmp_path / (name + ".s")
        f.write_text(src)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_kloop_asm.py"), str(f)], capture_output=True, text=True)
        assert r.returncode == want, (name, r.stdout)


def test_attention_assembly_is_what_the_generator_writes(tmp_path):
    """The committed iteration bodies of the opt-in assembly attention forward (attn_fwd32p_*.inc) are exactly what tools/gen_attn_asm.py
    emits; the common-case body has no scalar branch on the flags and the same MFMA count as one path of the generic body."""
    import subprocess
    import sys
    env = dict(os.environ, GEN_OUT_DIR=str(tmp_path))
    env.pop("GEN_ATTN_ABL", None)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_attn_asm.py")], check=True, env=env, capture_output=True)
    names = ["attn_fwd32p_tile0.inc", "attn_fwd32p_tile1.inc", "attn_fwd32p_tile0c.inc", "attn_fwd32p_tile1c.inc", "attn_fwd32p_clobbers.inc"]
    for n in names:
        new, old = open(tmp_path / n).read(), open(os.path.join(ROOT, "mla_amd", "csrc", n)).read()
        assert new == old, f"{n} differs from the generator's output: run python tools/gen_attn_asm.py"
    for par in (0, 1):
        body = [ln.strip('"\\n \n') for ln in open(tmp_path / f"attn_fwd32p_tile{par}c.inc") if ln.startswith('"')]
        assert sum(ln.startswith("v_mfma_f32_32x32x16_bf16") for ln in body) == 32          # 16 Q K^T + 16 P V
        assert not any(ln.startswith(("s_bitcmp", "s_cbranch_scc", "s_branch")) for ln in body)
        assert sum(ln.startswith("s_cbranch_vccz") for ln in body) == 2                     # the running-max decision (data-dependent)
        assert sum(ln.startswith("global_load_lds_dwordx4") for ln in body) == 8            # K of tile k + 2, V of tile k + 1
        # the score buffers alternate: parity p accumulates Q K^T into the OTHER buffer than the one its softmax reads
        qk = [ln for ln in body if ln.startswith("v_mfma") and ln.split()[1].startswith("v[")]
        assert all(ln.split()[1].startswith("v[96:" if par == 0 else "v[64:") or ln.split()[1].startswith("v[112:" if par == 0 else "v[80:") for ln in qk), qk[:2]


def test_attention_asm_checker_flags_persistent_registers_and_vacuous_input(tmp_path):
    """tools/check_attn_asm.py (run by build.sh): compiler code between the tile statements that names v64.. or an AGPR is a finding;
    statement-local temporaries (v40..63) are not; a listing without the kernel or without its tile statements is a failure, not a pass."""
    import subprocess
    import sys
    stmt = ";;#ASMSTART\n\tv_mfma_f32_32x32x16_bf16 a[0:15], a[64:67], v[64:67], a[0:15]\n;;#ASMEND\n"
    def kern(between):
        return ("_ZN12_GLOBAL__N_118attn_fwd32p_kernelENS_8AttnArgsE:\n\ts_load_dword s0, s[0:1], 0x0\n" + stmt + between + stmt +
                "\tv_accvgpr_read_b32 v70, a3\n\ts_endpgm\n.end_amdhsa_kernel\n")
    cases = {"clean": (kern("\tv_add_u32_e32 v41, v3, v5\n\ts_barrier\n"), 0),
             "score_register": (kern("\tv_mov_b32_e32 v64, v1\n"), 1),
             "score_tuple": (kern("\tglobal_load_dwordx4 v[60:63], v[2:3], off\n\tglobal_load_dwordx4 v[62:65], v[2:3], off\n"), 1),
             "agpr": (kern("\tv_accvgpr_write_b32 a5, v1\n"), 1),
             "one_statement_only": ("_ZN12_GLOBAL__N_118attn_fwd32p_kernelENS_8AttnArgsE:\n" + stmt + "\ts_endpgm\n.end_amdhsa_kernel\n", 1),
             "kernel_missing": ("_Z5otherv:\n\ts_endpgm\n.end_amdhsa_kernel\n", 1)}
    for name, (src, want) in cases.items():
        f = tmp_path / (name + ".s")
        f.write_text(src)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_attn_asm.py"), str(f)], capture_output=True, text=True)
        assert r.returncode == want, (name, r.stdout)
