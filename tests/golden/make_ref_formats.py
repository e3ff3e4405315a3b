"""Generates tests/golden/ref_formats.json with the reference's OWN writers (oracle/_ref/ref_formats, built by
`make -C oracle ref` from src/Config.cpp and src/Utils/Timer.hpp where they lie):
  config  for every script text of ref_config.json (the 62 shipped input/**/*.txt without their `script` line) plus a
          few texts with the tokens no shipped script uses: the bytes Config::saveToFile writes (config.txt echo,
          Config.cpp:209-302); the script name is the marker @SCRIPT@ (see oracle/ref_formats.cpp)
  info    info.txt (main.cpp:338-358 + Timer::print, Timer.hpp:58-68) for a few (nV, nT, steps, iterations, timings)
Run in the build container (needs /root/reference); the GPU box only reads the JSON."""
import json
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_formats")

EXTRA = {
    "extra/tokens1": "energy SNH\ntimeStepper DOT 1\ntime 3 0.01\nturnOffGravity\ntol 2\n1e-3\n2e-4\nview perspective\nzoom 2.5\n",
    "extra/tokens2": "energy FCR\ntimeStepper GSDD -7 300\nrestart some/status12\nwarmStart 0\nhandleRatio 0.25\n"
                     "appendStr run7\ndisableCout\nshape input a/b/c.msh\n",
    "extra/tokens3": "energy NH\ntimeStepper LBFGSH 9\nstiffness 123 0.3\ndensity 7\nsize 2.5\nrotateModel 1 0 0 -30\n"
                     "tuning 3\n0.5\n1e-7\n12345678\ninexactSolve 1\nresolution 37\n",
    "extra/tokens4": "energy FCR\ntimeStepper ADMM 0\ntimeIntegration BE\nshape cylinder\nstiffness 1e6 0.49\n"
                     "time 1e-3 1.5e-5\ndensity 1234567\n",
    "extra/tokens5": "energy FCR\ntimeStepper ADMMDD 12\nshape Sharkey\nview sideways\nwarmStart 1\n",
}

INFO = [
    # nV nT iterNum innerIterAmt descent, 14 step slots, 7 temp3 slots   (the pattern of `dot_hip --dump-formats`)
    dict(nV=0, nT=0, iterNum=7, inner=123, t=[12.5] + [0.001 * (k + 1) for k in range(14)] + [0.0] * 7),
    dict(nV=17315, nT=86058, iterNum=200, inner=5190,
         t=[1.04137, 0.0078125, 0.0351562, 0.0, 0.348633, 0.261963, 0.0302124, 0.0256958, 0.0318298, 0.0599365, 0.0595703,
            0.00012207, 1234.56789, 1e-9, 0.0] + [0.0] * 7),
]


def main():
    import sys
    sys.path.insert(0, ROOT)
    from tests.test_oracle_pin import reference_script_text
    with open(os.path.join(HERE, "ref_config.json")) as f:
        texts = {k: reference_script_text(k) for k in json.load(f)}
    texts.update(EXTRA)
    out = dict(config={}, info=[])
    with tempfile.TemporaryDirectory() as d:
        for rel, text in sorted(texts.items()):
            src, dst = os.path.join(d, "s.txt"), os.path.join(d, "c.txt")
            with open(src, "w") as f:
                f.write(text)
            subprocess.check_call([EXE, "config", src, dst], stdout=subprocess.DEVNULL)
            with open(dst) as f:
                # the shipped scripts' texts are not stored (the tests read them from the reference checkout); ours are
                out["config"][rel] = dict(text=text, echo=f.read()) if rel in EXTRA else dict(echo=f.read())
        for rec in INFO:
            dst = os.path.join(d, "i.txt")
            args = [str(rec["nV"]), str(rec["nT"]), str(rec["iterNum"]), str(rec["inner"])] + [repr(t) for t in rec["t"]]
            subprocess.check_call([EXE, "info", dst] + args)
            with open(dst) as f:
                out["info"].append(dict(rec, text=f.read()))
    with open(os.path.join(HERE, "ref_formats.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("ref_formats.json:", len(out["config"]), "config echoes,", len(out["info"]), "info.txt")


if __name__ == "__main__":
    main()
