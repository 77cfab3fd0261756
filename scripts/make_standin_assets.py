"""Write the stand-in asset tree of BASELINE config 1 (models/, atlases/, test_data/ in the layout of the reference's release tarballs) under
<outdir>:  python scripts/make_standin_assets.py <outdir>   -- then OAI_DATA_DIR=<outdir> python -m pytest tests/test_real_assets_gpu.py -m gpu.
See oai_analysis_2_amd.synth.write_standin_asset_tree."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oai_analysis_2_amd.synth import write_standin_asset_tree
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "build", "standin_assets")
print(out, write_standin_asset_tree(out, os.path.join(ROOT, "tests", "golden", "segment_small.npz")))
