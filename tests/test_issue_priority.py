"""csrc/issue_priority.py, the build step between the compiler's assembly of the strip kernel and the assembler: it may
only INSERT `s_setprio` pairs around runs of instructions that cannot share an issue turn -- never drop, reorder or touch an
instruction -- and the library in the tree must be the one built that way (no packed fp32 arithmetic in the brightness /
gradient strip kernels, priority changes present)."""
import importlib.util
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cuda-flow2d_amd", "csrc")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def load_filter():
    spec = importlib.util.spec_from_file_location("issue_priority", os.path.join(CSRC, "issue_priority.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


SNIPPET = """\t.text
\t.globl\tkernel
kernel:
\tv_mul_f32_e32 v1, v2, v3
\tv_sub_f32_dpp v4, v5, v6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
\tv_add_f32_e32 v7, v1, v4
\tv_sub_f32_dpp v8, v5, v6 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
\tv_add_f32_e32 v9, v1, v4
\tv_add_f32_e32 v9, v9, v4
\tv_add_f32_e32 v9, v9, v4
\tv_add_f32_e32 v9, v9, v4
\tv_rcp_f32_e32 v10, v9
\ts_cbranch_scc1 .LBB0_2
.LBB0_1:                                ; %loop
\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5]
.LBB0_2:
\tv_sqrt_f32_e32 v11, v9
\ts_endpgm
\t.section\t.rodata,"a",@progbits
\t.p2align\t6, 0x0
\t.amdhsa_kernel kernel
\tv_rcp_f32_e32 v10, v9
"""


def run_filter(tmp_path, text, gap):
    src, dst = tmp_path / "in.s", tmp_path / "out.s"
    src.write_text(text)
    p = subprocess.run(["python3", os.path.join(CSRC, "issue_priority.py"), str(gap), str(src), str(dst)], stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr
    return dst.read_text().splitlines(keepends=True)


@pytest.mark.parametrize("gap", [0, 1, 2, 6])
def test_filter_only_inserts_priority_pairs(tmp_path, gap):
    out = run_filter(tmp_path, SNIPPET, gap)
    kept = [l for l in out if l.strip() not in ("s_setprio 3", "s_setprio 0")]
    assert kept == SNIPPET.splitlines(keepends=True)  # every original line, in order, untouched
    ups = [i for i, l in enumerate(out) if l.strip() == "s_setprio 3"]
    downs = [i for i, l in enumerate(out) if l.strip() == "s_setprio 0"]
    assert len(ups) == len(downs) and all(u < d for u, d in zip(ups, downs))
    assert all(d < u2 for d, u2 in zip(downs, ups[1:]))  # never nested
    for u, d in zip(ups, downs):  # a run never spans a label, a directive or a branch
        body = out[u + 1:d]
        assert body and all(l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\ts_cbranch") for l in body)
    text = "".join(out)
    assert "\t.amdhsa_kernel kernel\n\tv_rcp_f32_e32 v10, v9\n" in text  # nothing outside the text section is touched
    # the two DPP subtractions are one run when a single plain instruction may sit between them, two runs otherwise
    first_block = text.split("v_add_f32_e32 v9, v1, v4")[0]
    assert first_block.count("s_setprio 3") == (2 if gap == 0 else 1)
    # the transcendental four plain instructions further on joins that run only for a gap of at least four
    assert text.split("s_cbranch_scc1")[0].count("s_setprio 3") == (first_block.count("s_setprio 3") + 1 if gap < 4 else 1)


def test_filter_classes():
    f = load_filter()
    special = ["v_pk_fma_f32", "v_pk_mov_b32", "v_mov_b64", "v_rcp_f32_e32", "v_rsq_f32_e32", "v_sqrt_f32_e32", "v_sub_f32_dpp",
               "v_mov_b32_dpp", "v_mov_b32_sdwa", "v_readfirstlane_b32", "v_cvt_pk_bf16_f32", "v_permlane32_swap_b32", "v_exp_f32_e32"]
    plain = ["v_fma_f32", "v_mul_f32_e32", "v_add_f32_e64", "v_cndmask_b32_e64", "v_lshl_add_u32", "v_min3_u32", "v_cmp_lt_f32_e32",
             "s_setprio", "global_load_dword", "ds_read_b64", "v_add_f64", "v_mov_b32_e32"]
    assert all(f.SPECIAL.search(op) for op in special)
    assert not any(f.SPECIAL.search(op) for op in plain)


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="no llvm-objdump in this image")
def test_the_library_in_the_tree_is_built_that_way(flow2d, tmp_path):
    lib = flow2d.HIP_LIB_PATH  # (the fixture has built it if this is a fresh checkout)
    assert os.path.exists(lib)
    work = tmp_path / "lib.so"
    os.symlink(lib, work)
    subprocess.run([OBJDUMP, "--offloading", str(work)], cwd=tmp_path, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    pipeline = lone = 0
    for name in sorted(os.listdir(tmp_path)):
        if "gfx950" not in name:
            continue
        dis = subprocess.run([OBJDUMP, "-d", str(tmp_path / name)], stdout=subprocess.PIPE, text=True, check=True).stdout
        kernels = [(k.split("\n", 1)[0], k) for k in re.split(r"\n(?=[0-9a-f]+ <)", dis) if re.search(r"fused_outer_kernelILi\d+ELi[0-3]E", k.split("\n", 1)[0])]
        if not kernels:
            continue
        # a code object holds one build of the kernel: the pipeline's (priority filter, no packed fp32 arithmetic outside the
        # log-derivative term) or the one for under-filled launches of a lone context (packed arithmetic, no filter: csrc/Makefile)
        filtered = ["s_setprio 3" in k for _, k in kernels]
        assert all(filtered) or not any(filtered), name
        for head, kernel in kernels:
            grad = re.search(r"fused_outer_kernelILi\d+ELi([0-3])E", head).group(1)
            if filtered[0]:
                pipeline += 1
                assert "s_setprio 0" in kernel, head
                if grad != "3":
                    assert "v_pk_" not in kernel, head
            else:
                lone += 1
                assert grad != "3" and "v_pk_" in kernel and "s_setprio" not in kernel, head
    assert pipeline >= 60 and lone >= 40  # 80 + 60 instantiations (developer builds hold fewer and are not what ships)
