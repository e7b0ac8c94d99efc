"""Development builds of the physics library with extra -D flags: python tools/build_variant.py NAME -DFLAG [-DFLAG...]  ->  build_exp/libmi_physics_NAME.so
(git-ignored; travels to the GPU box with the snapshot; selected at run time with MI_PHYSICS_LIB=build_exp/libmi_physics_NAME.so).  Several NAME:flags groups
separated by '--' are compiled in parallel."""
import subprocess
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from d3d12renderer_amd import build as hb   # noqa: E402


def main():
    groups, cur = [], []
    for a in sys.argv[1:]:
        if a == "--":
            groups.append(cur); cur = []
        else:
            cur.append(a)
    if cur:
        groups.append(cur)
    out = ROOT / "build_exp"; out.mkdir(exist_ok=True)
    procs = []
    for g in groups:
        name, flags = g[0], g[1:]
        lib = out / f"libmi_physics_{name}.so"
        cmd = [hb.hipcc(), *hb.FLAGS, *flags, "-I", str(ROOT / "include"), *map(str, hb.SOURCES), "-o", str(lib)]
        procs.append((name, lib, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    rc = 0
    for name, lib, p in procs:
        o, _ = p.communicate()
        print(name, "->", lib if p.returncode == 0 else "FAILED")
        if p.returncode:
            print(o[-4000:]); rc = 1
    sys.exit(rc)


if __name__ == "__main__":
    main()
