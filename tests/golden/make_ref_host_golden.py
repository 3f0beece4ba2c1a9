"""Generates tests/golden/ref_host_golden.npz by RUNNING THE REFERENCE'S OWN HOST CODE in this container.

oracle/_ref/ref_host_probe is oracle/ref_host_probe.cpp (ours, a command-line shell) linked with the
reference's host sources compiled where they lie under /root/reference (oracle/Makefile, target ref-host):
GetMaxWarpLevel, ComputeGaussianKernel, Data2D raw IO, OperationParameters, IOUtils' colour-wheel PPM and
magnitude writers, Settings::LoadSettings over the vendored TinyXML.  This script feeds them seeded inputs
and records inputs and outputs.  The fixture is data; tests/test_oracle.py and tests/test_host_cpu.py hold
the oracle and the product's host layer to it.

Run here (needs /root/reference for the build):   python tests/golden/make_ref_host_golden.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
PROBE = os.path.join(ROOT, "oracle", "_ref", "ref_host_probe")

LEVEL_SHAPES = [(584, 388, 0.9), (584, 388, 0.8), (128, 128, 0.9), (1024, 1024, 0.5), (1920, 1080, 0.5),
                (4096, 4096, 0.5), (8192, 8192, 0.5), (100, 70, 0.8), (131, 77, 0.95), (256, 192, 0.5), (64, 64, 0.75),
                (5, 5, 0.5), (4, 4, 0.9), (1000, 3, 0.5), (640, 480, 0.1), (640, 480, 0.01), (300, 200, 0.99),
                (100, 100, 1.0), (100, 100, 1.5)]
SIGMAS = [0.34, 0.45, 0.7, 1.0, 1.5, 1.9, 2.2, 3.0, 5.0, 8.3]

SETTINGS_XML = """<?xml version="1.0"?>
<!-- Settings for the Optical flow computation program -->
<OpticalFlow>
  <Input>
    <Path inputPath="./data/"/>
    <Mode Nx="584" Ny="388" imageType="8-bit">
    	<Files file1 ="rub1.raw" file2 ="rub2.raw"/>
    </Mode>
  </Input>
  <Parameters>
    <Method mode ="2d" run="flow" key="0" />
    <Solver>
      <Iterations inner="7" outer="13"/>
      <Warping levels="20" scaling="0.85" medianRadius="3"/>
      <Model sigma="0.45" alpha ="3.5" e_smooth="0.002" e_data="0.1"/>
    </Solver>
  </Parameters>
  <Output>
    <Path outputPath="./out/"/>
  </Output>
</OpticalFlow>
"""


def run(*args):
    return subprocess.run([PROBE] + [str(a) for a in args], check=True, capture_output=True, text=True).stdout


def parse_settings(text):
    d = {}
    for line in text.strip().splitlines():
        k, _, v = line.partition(" ")
        d[k] = v
    return d


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref-host"])
    out = {}
    meta = {}
    meta["levels"] = [[w, h, s, int(run("levels", w, h, repr(s)))] for w, h, s in LEVEL_SHAPES]
    meta["taps"] = {}
    for s in SIGMAS:
        fields = run("taps", repr(s)).split()
        meta["taps"][repr(s)] = {"radius": int(fields[0]), "bits": fields[1:]}
    meta["bag"] = [int(x) for x in run("bag").split()]
    rng = np.random.default_rng(77)
    with tempfile.TemporaryDirectory() as tmp:
        p = lambda name: os.path.join(tmp, name)
        # settings.xml: the reference's own file, and a variant with distinct values in every field
        meta["settings_reference_file"] = parse_settings(run("settings", "/root/reference/settings.xml"))
        open(p("s.xml"), "w").write(SETTINGS_XML)
        meta["settings_variant"] = parse_settings(run("settings", p("s.xml")))
        meta["settings_variant_xml"] = SETTINGS_XML
        meta["settings_missing_file"] = parse_settings(run("settings", p("absent.xml")))
        # raw readers / writers
        a8 = rng.integers(0, 256, (7, 9), dtype=np.uint8)
        a8.tofile(p("a8.raw"))
        run("readu8", p("a8.raw"), 9, 7, p("a8_f32.raw"))
        out["raw_u8_in"] = a8
        out["raw_u8_as_f32"] = np.fromfile(p("a8_f32.raw"), np.float32).reshape(7, 9)
        f = (rng.normal(100, 90, (7, 9))).astype(np.float32)
        f[0, :4] = [-3.7, 255.4, 255.6, 300.0]
        f.tofile(p("f.raw"))
        run("writeu8", p("f.raw"), 9, 7, p("f_u8.raw"))
        out["raw_f32_in"] = f
        out["raw_f32_as_u8"] = np.fromfile(p("f_u8.raw"), np.uint8).reshape(7, 9)
        # colour wheel + magnitude on a flow that visits every sector, the axes, zero and saturation
        h, w = 40, 48
        yy, xx = np.mgrid[0:h, 0:w]
        ang = (xx / w * 2 * np.pi).astype(np.float64)
        mag = (yy / (h - 1) * 14).astype(np.float64)
        u = (mag * np.cos(ang)).astype(np.float32)
        v = (mag * np.sin(ang)).astype(np.float32)
        u[0, :] = 0.0
        v[1, :] = 0.0
        u[2, :8] = [0.0, 1.0, 0.0, -1.0, 10.0, -10.0, 1e-3, 25.0]
        v[2, :8] = [0.0, 0.0, 1.0, 0.0, 10.0, -10.0, -1e-3, -0.5]
        u[3:6] += rng.normal(0, 1, (3, w)).astype(np.float32)
        u.tofile(p("u.raw"))
        v.tofile(p("v.raw"))
        run("ppm", p("u.raw"), p("v.raw"), w, h, "10.0", p("res.pgm"))
        run("amp", p("u.raw"), p("v.raw"), w, h, p("amp.raw"))
        out["flow_u"], out["flow_v"] = u, v
        out["ppm_bytes"] = np.frombuffer(open(p("res.pgm"), "rb").read(), np.uint8)
        out["amp"] = np.fromfile(p("amp.raw"), np.float32).reshape(h, w)
    out["meta"] = np.array(json.dumps(meta, indent=1))
    path = os.path.join(HERE, "ref_host_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    print(json.dumps(meta, indent=1)[:1500])


if __name__ == "__main__":
    sys.exit(main())
