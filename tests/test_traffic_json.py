"""profiles/traffic.json (the PMC figure bench.py prints as `roofline.traffic`) must have been measured on the dominant kernel's source
as it stands: every entry carries the sha256 of the kernel's region of its .hip file (tools/traffic_key.py).  Fails after an edit to that
kernel until `tools/pmc.sh` has been re-run on a GPU box and its fragment merged (`python tools/traffic_key.py merge ...`)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_traffic_entries_name_the_current_kernel_source():
    import traffic_key
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert "config2" in tj, "the headline configuration needs a PMC figure"
    for cfg, ent in tj.items():
        assert ent["read_bytes"] > 0 and ent["write_bytes"] > 0, cfg
        assert ent["kernel_region"] == traffic_key.KERNEL_OF_CONFIG[cfg], cfg
        assert ent["kernel_region_sha16"] == traffic_key.key_hash(ent["kernel_region"]), \
            "%s: %s changed since the PMC pass: re-run tools/pmc.sh and merge the fragment" % (cfg, ent["kernel_region"])


def test_region_hash_ignores_the_rest_of_the_file(tmp_path):
    import traffic_key
    src = open(os.path.join(ROOT, traffic_key.SOURCE)).read()
    d = tmp_path / "variational_mmt_amd" / "csrc"
    d.mkdir(parents=True)
    (d / "generator_fused.hip").write_text(src + "\n// an edit behind every region\n")
    assert traffic_key.key_hash("gen2p", root=str(tmp_path)) == traffic_key.key_hash("gen2p")
    (d / "generator_fused.hip").write_text(src.replace("struct G2P {", "struct G2P {  // touched", 1))
    assert traffic_key.key_hash("gen2p", root=str(tmp_path)) != traffic_key.key_hash("gen2p")
    assert traffic_key.key_hash("gen2w", root=str(tmp_path)) == traffic_key.key_hash("gen2w")
