"""Copies the round's artefacts from gpurun_out/round_<tag>/ (scripts/prof_round.sh <tag>) into profiles/ and writes profiles/<tag>_README.md.
usage: python scripts/make_profile_readme.py r02"""
import csv, glob, json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(R, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
O = os.path.join(R, "gpurun_out", f"round_{tag}")
line = [x for x in open(os.path.join(O, "bench_line.json")) if x.startswith("{")][-1]
open(os.path.join(P, f"{tag}_bench_line.json"), "w").write(line)
shutil.copy(glob.glob(os.path.join(O, "bench", "*kernel_stats.csv"))[0], os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
shutil.copy(os.path.join(O, "traffic_sres.json"), os.path.join(P, f"{tag}_pmc_traffic_sres.json"))
shutil.copy(os.path.join(O, "traffic_table.md"), os.path.join(P, f"{tag}_pmc_traffic_table.md"))
shutil.copy(os.path.join(O, "sq_summary.md"), os.path.join(P, f"{tag}_sq_summary.md"))
d = json.loads(line)
r, a, t = d["roofline"], d.get("fp32_mfma") or d.get("alt_precision"), json.load(open(os.path.join(P, f"{tag}_pmc_traffic_sres.json")))
rows = list(csv.DictReader(open(os.path.join(P, f"{tag}_bench_kernel_stats.csv"))))
n = sum(int(x["Calls"]) for x in rows if "conv3_igemm_sres" in x["Name"])
tt = sum(float(x["TotalDurationNs"]) for x in rows if "conv3_igemm_sres" in x["Name"])
par = d.get("parity") or {}
txt = f'''# profiles, round {tag[1:].lstrip("0")}

`{tag}_bench_kernel_stats.csv` : `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 1`
on one MI355X (wrapper: `scripts/prof_round.sh {tag}`).  `{tag}_bench_line.json` : the JSON line that same run printed.  The run
processes the default split-fp16 (fp16x3) arithmetic (1 warm-up + 3 timed volumes, + 2 segmentations for the `parity` block) and
then the same again in exact fp32 MFMA (the `fp32_mfma` object, full --steps), so both kernel families appear in the CSV.  All 160
tiles of a volume go through the U-Net in one pass ({d["config"]["tiles_per_pass"]} tiles per pass, 148 GiB of activation workspace).

Dominant kernel of the default mode = `conv3_igemm_sres<4, ...>` (split-resident fp16x3; all tile-shape instantiations):

* rocprofv3: {n} launches, average {tt / n / 1e6:.3f} ms
* bench.py HIP events (timed region, 3 volumes): {r["launches"]} launches, average {r["avg_launch_ms"]:.3f} ms
* achieved {r["achieved"]:.0f} TFLOP/s algorithmic (SURVEY 8d contract figure) = **{r["frac"]:.3f}** of the 2.5 PFLOP/s dense fp16 MFMA peak
  (round 1: 0.178); the kernel executes 3 MFMA passes per algorithmic product: executed {r["executed_frac"]:.2f} of nominal peak,
  {r["executed_frac_of_sustained_issue_rate"]:.2f} of the 1.857 PFLOP/s this chip sustains with operands in registers; phases of a wave's life and the
  ceiling argument: `{tag}_conv_phases.md`; SQ / GRBM counters of the same build: `{tag}_sq_summary.md`
* HBM-side traffic (`{tag}_pmc_traffic_sres.json`, separate `--pmc FETCH_SIZE` / `WRITE_SIZE` passes over a 32-tile pass, FETCH
  doubled per the guide; per-launch table `{tag}_pmc_traffic_table.md`): {t["fetch_x2_bytes_per_32_tile_pass"]/1e9:.1f} GB fetched + {t["write_bytes_per_32_tile_pass"]/1e9:.1f} GB written by the conv
  kernel's {t["launches_per_32_tile_pass"]} launches, {t["bytes_per_launch"]/1e9:.2f} GB per launch ({t["bytes_per_launch"]*5/1e9:.1f} GB per launch of the 160-tile pass); the counters include Infinity-Cache hits
* value **{d["value"]:.2f} volumes/s** ({d["ms_per_step"]:.1f} ms per volume: segment + register + 2 resamples; round 1: 5.06); the ICON registration of the
  same volume runs on a side stream underneath the segmentation
* full-size parity of that arithmetic against the reference's own CPU run (`tests/golden/segment_fullsize.npz`): {par.get("mask_flips")} mask flips of
  {par.get("mask_voxels")} voxels (max |p_ref - 0.5| at a flip {par.get("max_abs_pref_minus_half_at_flips", 0):.1e}), sum|dp| per 23.6 M voxels {[round(x, 2) for x in par.get("sum_abs_dp_per_23.6M_voxels", [])]} of the 12 the reference accepts

`conv3_igemm_f32` (exact fp32 MFMA, `fp32_mfma` in the line, {a["steps"] if "steps" in a else "?"} timed steps): {a["value"]:.2f} volumes/s, {a["roofline"]["achieved"]:.1f} TFLOP/s = {a["roofline"]["frac"]:.3f}
of the 157.3 TFLOP/s fp32 MFMA peak ({a["roofline"]["executed_frac"]:.2f} on the stricter frame-aware FLOP count).

Other files of the round: `{tag}_conv_phases.md` (s_memtime phase budget of the conv kernel, what changed, what was tried, the ceiling),
`{tag}_phase_stamps.txt` (raw output), `{tag}_packed_fp32_hazard.md` (wrong 16-lane groups from packed fp32 VALU ops beside the MFMA
kernels; why the library is built without them), `{tag}_registration.md` + `{tag}_registration_kernel_stats.csv` (config-3 warp loop op-by-op vs
fused chains, the fused two-map resample, one ICON direction with / without graph replay).
'''
open(os.path.join(P, f"{tag}_README.md"), "w").write(txt)
print(txt[:1500])
