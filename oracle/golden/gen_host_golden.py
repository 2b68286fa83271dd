#!/usr/bin/env python3
"""Regenerate tests/golden/host_uniforms.json and host_call_trace.json (TEST INFRASTRUCTURE, build container only).

Hands the settings JSON of the reference's own save (`saves/100 X 100 Test.weathersandbox`, decoded with codec.py) to
gen_host_golden.js, which executes slices of /root/reference/app.js against a recording mock GL context. TZ=UTC is required: the
reference's clock uses local-time Date accessors."""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("WX_REFERENCE", "/root/reference")


def main(out_dir=None):
    import wxpkg
    pkg = wxpkg.load_package()
    sf = pkg.codec.load(os.path.join(REF, "saves", "100 X 100 Test.weathersandbox"))
    out_dir = out_dir or os.path.join(ROOT, "tests", "golden")
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(sf.settings, f)
    try:
        subprocess.check_call(["node", os.path.join(HERE, "gen_host_golden.js"), "--out", out_dir, "--save-settings", f.name],
                              env=dict(os.environ, TZ="UTC", WX_REFERENCE=REF))
    finally:
        os.unlink(f.name)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
