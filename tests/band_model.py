"""Band-time model of one 64-column apply (round 4).  Predicts the duration of every launch from the PLAN alone and sets it
against a measured per-launch timeline; then prices alternative decompositions with the same constants.

  g++ -O2 -std=c++17 -pthread -I hifir_amd/csrc tests/cpp/plan_model.cpp -o /tmp/plan_model
  /tmp/plan_model hier.hifamd > plan.jsonl                    # (hier.hifamd: hifamd_save of the hierarchy; host only)
  python tests/launch_timeline.py gpurun_out/<trace> tl.txt   # (measured: rocprofv3 kernel trace of bench.py)
  python tests/band_model.py plan.jsonl [tl.txt] [--alt]

Constants (microseconds; all measured on one MI355X this round or the last):
  a dependent launch inside a replayed graph ...................... 1.8   (tests/microbench/launch_floor.hip)
  skeleton of a tile band (descriptor -> row ids -> right-hand sides -> LDS -> barrier -> product of one small component
  -> stores) ........................................................ 6.7   (bands with tiles and carried work switched off)
  streamed bytes of a band (rows x 1 KB in + out, inverse strips) .... 4.0 TB/s (level >= 1 leaf bands), 3.45 TB/s (level 0 L bands),
                                                                     4.7 TB/s (level 0 U bands: sinks streamed, two workgroups per unit)
  one wave's tile chain ............................................ 0.2 per tile of the band's longest wave, minus 2
  tile throughput, whole chip ...................................... 20 ns per tile and 16-column slice and unit (one slice per
                                                                     workgroup), 38 ns per tile and 32-column slice
  carried prefix (row walk by extra workgroups behind the band's) .. 2 + 30 ns per carried entry
  prefix pass in front of a top product ............................ 8 + 37 ns per entry
  top / tail product ............................................... 7 + 2 n^2 64 / 43 TFLOP/s, K-split reduction 5
  Schur product on tiles (levels >= 1) ............................. 6 + 25 ps per nonzero (64 columns)
  level 0: E product at 6.4 TB/s of its rows + gathers; output lists 3 + rows x 1 KB / 3.6 TB/s
"""
import json
import sys

LAUNCH, SKEL = 1.8, 6.7
BW_BAND, BW_L0, BW_L0_U = 4.0e6, 3.45e6, 4.7e6  # bytes per microsecond
TILE_CHAIN, TILE_CHAIN_OFF = 0.2, 2.0
TILE_THR1, TILE_THR2 = 0.020, 0.038
CARRY0, CARRY = 2.0, 3.0e-5
PREFIX0, PREFIX = 8.0, 3.7e-5
GEMM0, GEMM_TF, REDUCE = 7.0, 43e6, 5.0  # flop per microsecond
SPMM0, SPMM = 6.0, 2.5e-5  # per nonzero (64 columns): 40 nonzeros per nanosecond, the fabric's gather rate on 16-row tiles
CT_WIDE = 128


def load_plan(path):
    levels = []
    for ln in open(path):
        d = json.loads(ln)
        if "dense_n" in d:
            continue
        if "band" not in d:
            levels.append({"info": d, "L": [], "U": []})
        else:
            levels[-1][d["tri"]].append(d)
    return levels


def band_time(b, nxt, sparse_level0=False, tiles=True, carry=True):
    """one component band launch; nxt: the band whose carried prefix rides on it (or None)"""
    rows = b["rows"]
    if b.get("sparse"):  # level 0: sparse-own components, bandwidth
        extra = b["nnz"] * 12
        # (U bands stream their sinks -- k_band_us, two workgroups per unit -- and run at the rate a bare copy reaches)
        return LAUNCH + 5.0 + (rows * 1024 + extra) / (BW_L0_U if b.get("tri") == "U" else BW_L0)
    t = LAUNCH + SKEL + (rows * 1024 + b.get("inv_bytes", 0)) / BW_BAND
    if tiles and b.get("ct_tiles", 0) > 0:
        nct2 = b["wgs"] > CT_WIDE
        thr = b["ct_tiles"] * (2 * TILE_THR2 if nct2 else 4 * TILE_THR1) / 256.0
        t += max(max(0.0, TILE_CHAIN * b["ct_wave_max"] - TILE_CHAIN_OFF), thr)
    if carry and nxt is not None and nxt.get("fused") and nxt.get("carried", 0) > 0:
        t += CARRY0 + CARRY * nxt["carried"]
    return t


def top_time(nt, prefix_entries):
    return [("prefix", LAUNCH + PREFIX0 - LAUNCH + PREFIX * prefix_entries), ("top_gemm", GEMM0 + 2.0 * nt * nt * 64 / GEMM_TF),
            ("top_reduce", REDUCE)]


def ldu_times(lv, second, tiles=True, carry=True):
    out = []
    info = lv["info"]
    for tri in ("L", "U"):
        bands = lv[tri]
        for i, b in enumerate(bands):
            if b.get("top_band"):
                if tri == "L":
                    out += [(f"L{info['level']} {n}", t) for n, t in top_time(b["rows"], b["carried"])]
                continue
            nxt = bands[i + 1] if i + 1 < len(bands) and not bands[i + 1].get("top_band") else None
            t = band_time(b, nxt, tiles=tiles, carry=carry)
            if info["level"] == 0 and second and tri == "L":
                t += info["nnzF"] * 512 / 6.0e6 * (b["rows"] / max(1, info["m"]))  # fused F entries: one 512-byte gather each
            if info["level"] == 0 and second and tri == "U" and i == len(bands) - 1:
                t += 0.1 * b["rows"] * 1024 / BW_L0 / 2  # fused S7: scattered output rows
            out.append((f"L{info['level']} {tri}{b['band']}", t))
    return out


def apply_times(levels, tail_level, tiles=True, carry=True):
    """[(name, microseconds)] in launch order"""
    out = []

    def rec(l):
        lv = levels[l]
        info = lv["info"]
        if l >= tail_level:
            n = info["n"]
            out.append((f"tail n={n}", GEMM0 + 2.0 * n * n * 64 / GEMM_TF))
            out.append(("tail reduce", REDUCE))
            return
        out.extend(ldu_times(lv, False, tiles, carry))
        nm = info["n"] - info["m"]
        if nm > 0:
            if l == 0:
                out.append((f"L{l} E", LAUNCH + (nm * 1024 + info["nnzE"] * 512) / 6.4e6))
            else:
                out.append((f"L{l} E", SPMM0 + SPMM * info["nnzE"]))
            rec(l + 1)
            if l > 0:
                out.append((f"L{l} F", SPMM0 + SPMM * info["nnzF"]))
        out.extend(ldu_times(lv, True, tiles, carry))
        # output rows not written by the last U band: the child's rows and the other bands'
        ub = [b for b in lv["U"] if not b.get("top_band")]
        last_rows = ub[-1]["rows"] if ub else 0
        out.append((f"L{l} out", 3.0 + (info["n"] - last_rows) * 1024 / 3.6e6))

    rec(0)
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    levels = load_plan(args[0])
    # the tail operator: the first level of at most 4,096 rows and everything below it
    tail_level = next((i for i, lv in enumerate(levels) if lv["info"]["n"] <= 4096), len(levels))
    pred = apply_times(levels, tail_level)
    meas = None
    if len(args) > 1:
        meas = [float(ln.split()[7]) for ln in open(args[1]) if ln.strip() and not ln.startswith("#")]
    tot_p = sum(t for _, t in pred)
    print(f"# launches predicted: {len(pred)}" + (f", measured: {len(meas)}" if meas else ""))
    if meas and len(meas) == len(pred):
        print("# idx  launch                 predicted  measured  ratio")
        for i, ((n, t), m) in enumerate(zip(pred, meas)):
            print(f"{i:4d}  {n:22s} {t:9.1f} {m:9.1f} {t / m:6.2f}")
        tot_m = sum(meas)
        err = [abs(t - m) for (_, t), m in zip(pred, meas)]
        print(f"# total predicted {tot_p:.0f} us, measured {tot_m:.0f} us ({100 * (tot_p - tot_m) / tot_m:+.1f} %); mean |error| per launch "
              f"{sum(err) / len(err):.1f} us; launches within 25 %: {sum(1 for (_, t), m in zip(pred, meas) if abs(t - m) <= 0.25 * m)} of {len(meas)}")
    else:
        for i, (n, t) in enumerate(pred):
            print(f"{i:4d}  {n:22s} {t:9.1f}")
        print(f"# total predicted {tot_p:.0f} us")
    if "--alt" in sys.argv:
        alternatives(levels, tail_level, tot_p)


def alternatives(levels, tail_level, base):
    print("\n# ---- alternatives, priced with the same constants (microseconds per 64-column apply) ----")
    print(f"today's plan ......................................................... {base:7.0f}")
    # floor of THIS launch structure: every band at skeleton + streamed bytes (tiles and carried work free)
    fl = sum(t for _, t in apply_times(levels, tail_level, tiles=False, carry=False))
    print(f"same launches, tile phase and carried prefixes free (structure floor) . {fl:7.0f}")
    # (A) passes in SpMM form: a band's outside entries in a chip-wide tile kernel of its own (perfectly balanced, no chain),
    # the band kernel keeps right-hand sides -> product -> stores
    a = 0.0
    for l, lv in enumerate(levels[:tail_level]):
        for tri in ("L", "U"):
            for b in lv[tri]:
                if b.get("cd") and not b.get("sparse") and (b.get("ct_tiles", 0) > 0 or b.get("carried", 0) > 0):
                    tiles = b.get("ct_tiles", 0) + b.get("carried", 0) / 11.0
                    a += 2 * (LAUNCH + SKEL - 2.0 + tiles * 4 * TILE_THR1 / 256.0)  # one more launch per band and solve
    print(f"(A) outside entries of every band as a chip-wide tile launch of its own  {fl + a:7.0f}   (+{a:.0f} over the floor: the extra "
          f"dependent launch costs what the chain costs)")
    # (B) larger combined top operators (dense G of n_t rows): the passes they absorb against the product's n^2
    for nt_new in (6144, 8192, 16384):
        d = 0.0
        for lv in levels[1:tail_level]:
            info = lv["info"]
            if not info.get("top_n"):
                continue
            nt = info["top_n"]
            absorbed, rows = 0.0, nt
            bl = [b for b in lv["L"] if not b.get("top_band")]
            bu = [b for b in lv["U"] if not b.get("top_band")]
            # L's last bands and U's first ones, whole bands, while the top stays within nt_new rows
            for b, bu_ in zip(reversed(bl), bu):
                if rows + b["rows"] > nt_new:
                    break
                rows += b["rows"]
                absorbed += band_time(b, None) + band_time(bu_, None)
            d += 2 * ((2.0 * rows * rows * 64 - 2.0 * nt * nt * 64) / GEMM_TF - absorbed)
        print(f"(B) combined top operators of up to {nt_new:5d} rows ............................ {base + d:7.0f}   ({d:+.0f})")
    # (B') block-triangular (two-level) top: the absorbed rows as a second diagonal block, 3/4 of the dense product's flops
    for nt_new in (8192,):
        d = 0.0
        for lv in levels[1:tail_level]:
            info = lv["info"]
            if not info.get("top_n"):
                continue
            nt = info["top_n"]
            absorbed, rows = 0.0, nt
            bl = [b for b in lv["L"] if not b.get("top_band")]
            bu = [b for b in lv["U"] if not b.get("top_band")]
            for b, bu_ in zip(reversed(bl), bu):
                if rows + b["rows"] > nt_new:
                    break
                rows += b["rows"]
                absorbed += band_time(b, None) + band_time(bu_, None)
            extra = rows - nt
            flops = 2.0 * 64 * (nt * nt + extra * extra + 2 * nt * extra * 0.5)
            d += 2 * ((flops - 2.0 * nt * nt * 64) / GEMM_TF + GEMM0 + LAUNCH - absorbed)
        print(f"(B') two diagonal blocks + one coupling block, {nt_new} rows ..................... {base + d:7.0f}   ({d:+.0f})")
    # (C) levels 3 .. 5 (or 4 .. 5) under one dense operator
    for l0 in (3, 4):
        if l0 < len(levels):
            n = levels[l0]["info"]["n"]
            cur = sum(t for nme, t in apply_times(levels, tail_level) if nme.startswith(tuple(f"L{q} " for q in range(l0, len(levels)))) or nme.startswith("tail"))
            new = GEMM0 + 2.0 * n * n * 64 / GEMM_TF + REDUCE
            print(f"(C) levels {l0} .. {len(levels) - 1} ({n} rows) as ONE dense operator ............................ {base - cur + new:7.0f}   ({new - cur:+.0f}; "
                  f"{n * n * 8 / 1e9:.1f} GB)")
    # (D) what 40 % of 8 TB/s needs
    print("(D) 40 % of 8 TB/s on 7.65 GB ........................................... 2390")


if __name__ == "__main__":
    main()
