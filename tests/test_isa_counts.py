"""Build-time check of the hand-kept instruction counts behind the counted `s_waitcnt vmcnt(n)` of the row passes (no GPU).

The LDS-DMA prefetch of a row is retired by `hadi_wait_vmcnt(n)` with n = the number of vector-memory instructions the
wavefront issued AFTER that row's DMA: the DMA pieces of younger rows (`fetch()` returns their count) and the result stores
of the row steps in between (`hadi_put_block_stores<B, T>()`).  Vector-memory operations retire in issue order, so n may be
a LOWER bound of what was really issued (the wait is then stricter than needed) but never more: an over-count lets the wait
pass with a DMA piece still in flight and the next step reads a stale ring row -- silently, and only when the memory is slow
(the fp32-state kernels carried exactly that over-count until round 2: B/2 counted, B/4 quad stores emitted).  The emulator
does not model vmcnt and the strict-build comparison on the GPU only catches it if the timing cooperates; this test reads
the compiler's own assembly instead (hipcc --save-temps, as tools/kernel_regs.py does)."""
import os, re, subprocess, sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

PAD = lambda B, es: 8 if B < 4 else (32 if es == 4 else 16)                     # HADI_ROW_PAD
STORES = lambda B, es: 1 if B == 1 else (B // 4 if (es == 4 and B >= 4) else B // 2)  # hadi_put_block_stores
STRIP_NS = lambda B, G, es: (3 if es == 8 else 4) if G == 2 else 4              # HADI_STRIP_NS (default build)


@pytest.fixture(scope="module")
def kernels():
    import kernel_regs
    rows, asm = kernel_regs.collect()
    out = {"__rows__": rows}
    # (a kernel's text runs to its .Lfunc_end label: an early exit -- e.g. the strips' "this instance has fewer time steps" -- puts an
    # s_endpgm in the middle)
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)^\.Lfunc_end\d+:", asm, re.S | re.M):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        if name.startswith("void hadi_pass_a"):
            out[name.replace("(HadiSweepArgs, int)", "").replace("void ", "")] = m.group(2)
        out.setdefault("__all__", {})[name] = m.group(2)
    assert len(out) > 31
    return out


def test_python_mirrors_match_the_source():
    """The formulas above are mirrors of hadi_put_block_stores / HADI_ROW_PAD / HADI_STRIP_NS: tie them to the source text."""
    csrc = os.path.join(ROOT, "pde_based_heston_solver_gpu_accelerated_amd", "csrc")
    k = "".join(open(os.path.join(csrc, f)).read() for f in sorted(os.listdir(csrc)) if f.startswith("hadi_k"))
    c = open(os.path.join(csrc, "hadi_core.h")).read()
    assert "return B == 1 ? 1 : (sizeof(T) == 4 && B >= 4) ? B / 4 : B / 2;" in k
    assert "constexpr int NST = (MODE == 3 ? 0 : MODE == 1 ? 3 : 1) * hadi_put_block_stores<B, T>();" in k      # strips (Craig-Sneyd predictor: + R1, C2; table build: none counted)
    assert "for (int k = 0; k < NA; k++) aft[k] += NST;" in k
    assert "for (int k = 0; k < PD; k++) ya[k] += hadi_put_block_stores<B, T>();" in k       # shared ring
    assert "#define HADI_ROW_PAD(B, ES) ((B) < 4 ? 8 : ((ES) == 4 ? 32 : 16))" in c
    assert "#define HADI_STRIP_NS(B, G, ES) ((G) == 2 ? ((ES) == 8 ? 3 : 4) : ((B) <= 4 ? HADI_STRIP_NS_NARROW : 4))" in c
    assert "#define HADI_STRIP_NS_NARROW 4" in c


def _args(name):
    return [a.strip() for a in name[name.index("<") + 1:name.rindex(">")].split(",")]


def test_strip_kernels_issue_at_least_the_stores_and_exactly_the_dma_pieces_the_waits_count(kernels):
    seen = 0
    for name, body in kernels.items():
        if not name.startswith("hadi_pass_a_strip<"):
            continue
        B, amer, T, G, mode = _args(name)
        B, G, es, mode = int(B), int(G), 4 if T == "float" else 8, int(mode)
        n_dma = len(re.findall(r"\bglobal_load_lds_dwordx4\b", body))
        n_st = len(re.findall(r"\bglobal_store_dwordx4\b", body)) + (len(re.findall(r"\bglobal_store_dwordx2\b", body)) if (B == 2 and es == 4) else 0)
        # fetch(): NS + 1 call sites (the strip's first row and the rows 1 .. NS - 1 ahead in the prologue, one in the loop);
        # the P representation at 8 nodes per lane keeps one slot behind the prefetch: rows 0 .. NS - 2 ahead, NS sites
        keep = amer == "2" and B >= 8 and G == 1
        sites = STRIP_NS(B, G, es) + (0 if keep else 1)
        rowp = 64 * B * G + PAD(B, es)
        if G == 1:
            pieces = -(-(rowp * es // 16) // 64)          # hadi_row_dma_count: whole 1 KiB pieces + the partial one
        else:
            pieces = (B * es // 16) + 1                   # hadi_half_row_to_lds: the half's pieces + the pad piece (low half)
        assert n_dma == sites * pieces, (name, n_dma)
        # hadi_strip_step is instantiated twice (last v-row or not): each copy stores the row block once; the Craig-Sneyd
        # predictor (mode 1) stores R1 and C2 as well, the corrector (mode 2) has one copy of the step
        # (mode 3 -- the table of the pairs' coupling column -- stores one double per lane and nothing the waits count)
        assert n_st >= (6 if mode == 1 else 1 if mode == 2 else 0 if mode == 3 else 2) * STORES(B, es), (name, n_st)
        if mode == 2:
            # the corrector's R1 / C2 rows: ordinary register loads of the compiler, B / 2 per row and array at two sites
            # (prologue, loop) -- the counted waits of the DMA ring add them as a lower bound -- plus the prologue's three
            # register rows of a strip without a partner
            n_x4 = len(re.findall(r"\bglobal_load_dwordx4\b", body))
            assert n_x4 >= 2 * 2 * (B // 2) + 3 * (B // 2), (name, n_x4)
        seen += 1
    assert seen >= 20


def test_ring_kernels_issue_at_least_the_stores_the_waits_count(kernels):
    seen = 0
    for name, body in kernels.items():
        if not name.startswith("hadi_pass_a<"):
            continue
        B, G, W, NG, PD, amer, mode, T = _args(name)
        B, es = int(B), 4 if T == "float" else 8
        wide = len(re.findall(r"\bglobal_store_dwordx4\b", body))
        narrow = len(re.findall(r"\bglobal_store_dwordx2\b", body)) + len(re.findall(r"\bglobal_store_dword\b", body))
        n_st = wide if B >= 2 and not (B == 2 and es == 4) else narrow   # one node per lane stores single elements
        assert n_st >= 2 * STORES(B, es), (name, n_st)   # two copies of hadi_row_step (last v-row or not)
        seen += 1
    assert seen >= 25


def test_pair_strip_kernels_issue_the_dma_pieces_and_stores_the_waits_count(kernels):
    """hadi_pass_a_pairs: HADI_PAIR_DMA = 6 LDS-DMA instructions per fetch (four 1 KiB pieces of both rows + the two 128-byte
    tails), five fetch sites (the first row and the rows 1 .. 3 ahead in the prologue, one in the loop; the P representation
    keeps the first row's slot and has the rows 1 .. 2 ahead: four), HADI_PAIR_STORES = 4 counted row stores per step (one
    copy of the step)."""
    csrc = os.path.join(ROOT, "pde_based_heston_solver_gpu_accelerated_amd", "csrc")
    k = "".join(open(os.path.join(csrc, f)).read() for f in sorted(os.listdir(csrc)) if f.startswith("hadi_k"))
    assert "#define HADI_PAIR_DMA 6" in k and "#define HADI_PAIR_STORES 4" in k
    seen = 0
    for name, body in kernels.items():
        if not name.startswith("hadi_pass_a_pairs<"):
            continue
        n_dma = len(re.findall(r"\bglobal_load_lds_dwordx4\b", body))
        n_st = len(re.findall(r"\bglobal_store_dwordx4\b", body))
        assert n_dma == (4 if _args(name)[0] == "2" else 5) * 6, (name, n_dma)
        assert n_st >= 4, (name, n_st)
        seen += 1
    assert seen == 3


def test_no_row_kernel_with_counted_waits_touches_scratch(kernels):
    """The counted `s_waitcnt vmcnt(n)` of the strip and pair-strip kernels know the LDS-DMA pieces and the result stores of a
    row step and nothing else: a register spilled into scratch is reloaded by a vector-memory operation inside the row loop
    that the count does not include -- the wait would pass with a DMA piece still in flight (and, at best, the reload drains
    the prefetch).  No such kernel may have a scratch segment or a spilled VGPR (compiler metadata, tools/kernel_regs.py)."""
    seen = 0
    for name, vgpr, sgpr, spills, scratch, lds in kernels["__rows__"]:
        if "hadi_pass_a_strip<" in name or "hadi_pass_a_pairs<" in name:
            assert spills == 0 and scratch == 0, (name, vgpr, spills, scratch)
            assert vgpr <= 256
            seen += 1
    assert seen >= 17


def _sregs(text):
    """Scalar registers an instruction's text names: s12, s[12:19], vcc / exec / m0 are not data registers of ours."""
    regs = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(a) for a in re.findall(r"\bs(\d+)\b", text))
    return regs


def test_row_table_entries_in_flight_are_not_touched_before_their_wait(kernels):
    """hadi_sload_issue requests a row-table entry with three `s_load_dwordx8` from inline asm and hadi_sload_wait retires them
    (`s_waitcnt lgkmcnt(0)`, the octets tied to it as operands).  Between the two the destination registers hold a load IN
    FLIGHT -- values the compiler believes ready.  If it ever copied or parked them there (live-range splitting, SGPR pressure:
    v_writelane spills), the copy would be stale; that is exactly what happened to the VECTOR registers of round 4's first
    Craig-Sneyd corrector.  So: in every kernel, no instruction between one of our load triples and the
    first lgkmcnt(0) wait behind it (any such wait retires them) may name a register of the three octets -- on EVERY path: the
    code in between branches (the vmcnt switch, the conditional fetch, the back edge of a rotated loop), so the test walks the
    control flow from the triple."""
    seen = 0
    for name, body in kernels["__all__"].items():
        lines = [l.split(";")[0].strip() for l in body.split("\n")]
        lines = [l for l in lines if l and (l.endswith(":") or not l.startswith("."))]
        label = {l[:-1]: k for k, l in enumerate(lines) if l.endswith(":")}
        for i in range(len(lines) - 2):
            trip = lines[i:i + 3]
            if not (all(t.startswith("s_load_dwordx8") for t in trip) and [t.rsplit(",", 1)[1].strip() for t in trip] == ["0x0", "0x20", "0x40"]):
                continue
            dst = set()
            for t in trip:
                dst |= _sregs(t.split(",")[0])
            assert len(dst) == 24, (name, trip)
            # every path from behind the triple to the first lgkmcnt(0) wait (forward branches, and the back edge of a rotated loop)
            todo, visited = [i + 3], set()
            while todo:
                k = todo.pop()
                while k not in visited:
                    visited.add(k)
                    assert k < len(lines), name
                    ln = lines[k]
                    if re.search(r"s_waitcnt\b.*lgkmcnt\(0\)", ln):
                        break
                    assert not ln.startswith("s_endpgm"), (name, "a path leaves the kernel with the loads in flight")
                    if not ln.endswith(":"):
                        assert not (_sregs(ln) & dst), (name, ln, sorted(dst))
                    m = re.match(r"s_(c?branch)\w*\s+(\S+)", ln)
                    if m:
                        todo.append(label[m.group(2)])
                        if m.group(1) == "branch":
                            break
                    k += 1
            seen += 1
    assert seen >= 25


def test_pair_strip_kernels_issue_the_dma_pieces_and_stores_the_waits_count(kernels):
    """hadi_pass_a_pairs: HADI_PAIR_DMA = 6 LDS-DMA instructions per fetch (four 1 KiB pieces of both rows + the two 128-byte
    tails), five fetch sites (the first row and the rows 1 .. 3 ahead in the prologue, one in the loop; the P representation
    keeps the first row's slot and has the rows 1 .. 2 ahead: four), HADI_PAIR_STORES = 4 counted row stores per step (one
    copy of the step)."""
    csrc = os.path.join(ROOT, "pde_based_heston_solver_gpu_accelerated_amd", "csrc")
    k = "".join(open(os.path.join(csrc, f)).read() for f in sorted(os.listdir(csrc)) if f.startswith("hadi_k"))
    assert "#define HADI_PAIR_DMA 6" in k and "#define HADI_PAIR_STORES 4" in k
    seen = 0
    for name, body in kernels.items():
        if not name.startswith("hadi_pass_a_pairs<"):
            continue
        n_dma = len(re.findall(r"\bglobal_load_lds_dwordx4\b", body))
        n_st = len(re.findall(r"\bglobal_store_dwordx4\b", body))
        assert n_dma == (4 if _args(name)[0] == "2" else 5) * 6, (name, n_dma)
        assert n_st >= 4, (name, n_st)
        seen += 1
    assert seen == 3


def test_no_row_kernel_with_counted_waits_touches_scratch(kernels):
    """The counted `s_waitcnt vmcnt(n)` of the strip and pair-strip kernels know the LDS-DMA pieces and the result stores of a
    row step and nothing else: a register spilled into scratch is reloaded by a vector-memory operation inside the row loop
    that the count does not include -- the wait would pass with a DMA piece still in flight (and, at best, the reload drains
    the prefetch).  No such kernel may have a scratch segment or a spilled VGPR (compiler metadata, tools/kernel_regs.py)."""
    seen = 0
    for name, vgpr, sgpr, spills, scratch, lds in kernels["__rows__"]:
        if "hadi_pass_a_strip<" in name or "hadi_pass_a_pairs<" in name:
            assert spills == 0 and scratch == 0, (name, vgpr, spills, scratch)
            assert vgpr <= 256
            seen += 1
    assert seen >= 17
