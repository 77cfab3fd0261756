"""Regenerates profiles/r01_README.md and the split-resident section of profiles/r01_pmc_traffic.md from the artefacts that
scripts/prof_round.sh produced (copied into profiles/)."""
import csv, json, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(R, "profiles")
d = json.loads(open(os.path.join(P, "r01_bench_line.json")).read())
r, a, t = d["roofline"], d["alt_precision"], json.load(open(os.path.join(P, "r01_pmc_traffic_sres.json")))
rows = list(csv.DictReader(open(os.path.join(P, "r01_bench_kernel_stats.csv"))))
n = sum(int(x["Calls"]) for x in rows if "conv3_igemm_sres" in x["Name"])
tt = sum(float(x["TotalDurationNs"]) for x in rows if "conv3_igemm_sres" in x["Name"])
txt = f'''# profiles, round 1

`r01_bench_kernel_stats.csv` : `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 1`
on one MI355X (wrapper: `scripts/prof_round.sh`).  `r01_bench_line.json` : the JSON line that same run printed.  The run
processes 4 volumes with the default split-fp16 conv arithmetic (1 warm-up + 3 timed) and then 3 volumes in exact fp32 MFMA
(1 warm-up + 2: the `alt_precision` object), so both kernel families appear in the CSV.  All 160 tiles of a volume go through
the U-Net in one pass ({d["config"]["tiles_per_pass"]} tiles per pass, 148 GiB of activation workspace).

Dominant kernel of the default mode = `conv3_igemm_sres<4, ...>` (split-resident fp16x3; all tile-shape instantiations):

* rocprofv3: {n} launches, average {tt / n / 1e6:.3f} ms
* bench.py HIP events (timed region, 3 volumes): {r["launches"]} launches, average {r["avg_launch_ms"]:.3f} ms
* achieved {r["achieved"]:.0f} TFLOP/s algorithmic (SURVEY 8d contract figure) = {r["frac"]:.3f} of the 2.5 PFLOP/s dense fp16 MFMA peak;
  the kernel executes 3 MFMA passes per algorithmic product: executed {r["executed_frac"]:.2f} of nominal peak,
  {r["executed_frac_of_sustained_issue_rate"]:.2f} of the 1.857 PFLOP/s this chip sustains with operands in registers (`r01_ablation.md`);
  the path is power-limited (1.96 GHz at 1.28 kW, `r01_power.md`), SQ counters in `r01_pmc_sres.md`
* HBM-side traffic (`r01_pmc_traffic_sres.json`, separate `--pmc FETCH_SIZE` / `WRITE_SIZE` passes over a 32-tile pass, FETCH
  doubled per the guide): {t["fetch_x2_bytes_per_32_tile_pass"]/1e9:.0f} GB fetched + {t["write_bytes_per_32_tile_pass"]/1e9:.1f} GB written by the conv kernel's {t["launches_per_32_tile_pass"]} launches,
  {t["bytes_per_launch"]/1e9:.2f} GB per launch ({r["traffic"]/1e9:.1f} GB per launch of the 160-tile pass); the counters include Infinity-Cache hits.
  History: 77 GB fetched with chunk-interleaved records and launch-order blocks; chunk-planar records (whole 128-B lines per
  DMA) and dealing the logical block list to the XCDs in groups of 32 (halo-sharing neighbours and cout blocks on one L2) brought
  it to {t["fetch_x2_bytes_per_32_tile_pass"]/1e9:.0f} GB -- worth only ~1 % of time: the DMA is asynchronous and the data was coming from the Infinity Cache.
* value {d["value"]:.2f} volumes/s ({d["ms_per_step"]:.1f} ms per volume: segment + register + 2 resamples); the ICON registration of the same volume runs on a side stream underneath the segmentation; box-to-box spread of this build: 5.03-5.13

`conv3_igemm_f32` (alt precision f32, exact fp32 MFMA): {a["value"]:.2f} volumes/s, {a["roofline"]["achieved"]:.1f} TFLOP/s = {a["roofline"]["frac"]:.3f} of the
157.3 TFLOP/s fp32 MFMA peak ({a["roofline"]["executed_frac"]:.2f} on the stricter frame-aware FLOP count).

Other files: `r01_pmc_traffic.md/.json` (HBM-side bytes of the fp32 kernels + the split-resident table), `r01_ablation.md` (what
each ingredient of the conv kernels costs + the chip's sustained MFMA rates), `r01_power.md` (clock / power while the paths run),
`r01_pmc_sres.md` (SQ counters of the default conv kernel), `r01_warp_160.md` + `r01_warp_160_kernel_stats.csv` (grid_sample /
compose at 160^3 against the HBM roofline), `r01_mesh.md` (marching cubes / smoothing / distance at full size),
`r01_initial_*` (the first working version of the round: 1.18 volumes/s).
'''
open(os.path.join(P, "r01_README.md"), "w").write(txt)
tab = os.path.join(R, "gpurun_out", "round", "traffic_table.md")
if os.path.exists(tab):
    p = os.path.join(P, "r01_pmc_traffic.md")
    s = open(p).read()
    i = s.index("\n## Split-resident fp16x3 path")
    s = s[:i] + "\n## Split-resident fp16x3 path (default), same method, warm 32-tile pass of `PREC=fp16x3 python3 scripts/perf_layers.py` (scripts/prof_round.sh)\n\n" + open(tab).read()
    open(p, "w").write(s)
print(txt[:600])
