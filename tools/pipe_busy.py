"""From the PMC summaries of tools/prof_small_pmc.sh: how busy the MFMA / f64-VALU pipe of a SIMD is during the fused kernel.

    python tools/pipe_busy.py profiles/r05_pmc_C2.json [...]        python tools/pipe_busy.py --gibbs profiles/r05_gibbs_pmc.json

MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) (busy cycles are counted per SIMD, GUI_ACTIVE
per XCD).  The f64 VALU shares that pipe: SQ_ACTIVE_INST_VALU counts quad-cycles in which a wave issues VALU work (MFMAs
excluded: their issue is one quad-cycle each), x 4 = cycles.  SQ_WAIT_ANY (parked at s_waitcnt / barrier), SQ_WAIT_INST_ANY
(issue stall: pipe or dependency) and the active counters add up to SQ_WAVE_CYCLES (MI355X_MICROARCH.md)."""
import json, sys

args = [a for a in sys.argv[1:] if not a.startswith('--')]
gibbs = '--gibbs' in sys.argv
for f in args:
    d = json.load(open(f))
    for k, v in sorted(d.items()):
        if not (('k_fused' in k and not gibbs) or ('k_gibbs_rate' in k and gibbs)):
            continue
        g = lambda n: v[n]['avg'] if n in v else float('nan')
        simd_cycles = g('GRBM_GUI_ACTIVE') / 8 * 1024
        if gibbs:
            print("%s  %s" % (f.split('/')[-1], k[:50]))
            print("   VALU instructions %.3g (%.1f per 64 rate evaluations at 128 x 11 x 600 000), SALU %.3g" %
                  (g('SQ_INSTS_VALU'), g('SQ_INSTS_VALU') / (128 * 11 * 600000 / 64.0), g('SQ_INSTS_SALU')))
            print("   VALU-active share of the SIMD cycles %.2f; LDS bank-conflict cycles / LDS-active cycles %.3f"
                  % (4 * g('SQ_ACTIVE_INST_VALU') / simd_cycles, g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE')))
            continue
        mf = g('SQ_VALU_MFMA_BUSY_CYCLES') / simd_cycles
        va = 4 * g('SQ_ACTIVE_INST_VALU') / simd_cycles
        wc = g('SQ_WAVE_CYCLES')
        print("%s  %s" % (f.split('/')[-1], k[:50]))
        print("   MFMA busy %.3f + f64-VALU active %.3f = pipe %.3f of the SIMD cycles; VALU / MFMA instructions %.1f"
              % (mf, va, mf + va, g('SQ_INSTS_VALU') / g('SQ_INSTS_MFMA')))
        print("   wave cycles: parked (waitcnt / barrier) %.2f, issue stall %.2f, VALU issue %.2f; HBM fetch %.1f MB write %.1f MB"
              % (g('SQ_WAIT_ANY') / wc, g('SQ_WAIT_INST_ANY') / wc, g('SQ_ACTIVE_INST_VALU') / wc, 2 * g('FETCH_SIZE') / 1024,
                 g('WRITE_SIZE') / 1024))
