"""NF kernel variants on the GPU box: python tools/nf_sweep.py  -- runs tools/nf_shapes.py against the default library
and the experimental builds under gl-abc-mcmc_amd/csrc/exp/ (GLABC_HIP_LIB), with and without forced tile mode."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = [("w8", None, None)]
for w in (12, 16):
    lib = os.path.join(root, "gl-abc-mcmc_amd", "csrc", "exp", "libglabc_hip_w%d.so" % w)
    if os.path.exists(lib):
        variants.append(("w%d" % w, lib, None))
        variants.append(("w%d-tile%d" % (w, w), lib, str(w)))
variants.append(("w8-tile8", None, "8"))
for name, lib, tile in variants:
    env = dict(os.environ)
    if lib:
        env["GLABC_HIP_LIB"] = lib
    if tile:
        env["GLABC_NF_TILE_WAVES"] = tile
    print("==== %s" % name, flush=True)
    subprocess.run([sys.executable, os.path.join(root, "tools", "nf_shapes.py")], env=env, cwd=root)
