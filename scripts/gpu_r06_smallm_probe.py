"""Round-6 probe: where is the floor of the small-M pointwise GEMMs of config 3's layer3 / layer4?

For each problem every fp32 tile x split-K factor is timed (back-to-back launches of the same kernel, ten per sample) and
the best six are printed, next to a near-empty launch (M = 32, K = 4) = the launch floor of the box.

    python scripts/gpu_r06_smallm_probe.py
"""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "scripts")
import conv_micro as cm  # noqa: E402

lib = cm.lib
PROBLEMS = [
    ("floor (M=32, K=4, N=4)", "1,1,4,8,4,4,1,1,0"),
    ("layer3 conv1.spatial  1024->204", "8,4,7,7,1024,204,1,1,0"),
    ("layer3 conv1.temporal 204->256", "8,4,7,7,204,256,1,1,0"),
    ("layer3 conv3.spatial  256->204", "8,4,7,7,256,204,1,1,0"),
    ("layer3 conv3.temporal 204->1024 +res", "8,4,7,7,204,1024,1,1,0,res"),
    ("layer4 conv1.spatial  2048->409", "8,2,4,4,2048,409,1,1,0"),
    ("layer4 conv1.temporal 409->512", "8,2,4,4,409,512,1,1,0"),
    ("layer4 conv3.spatial  512->409", "8,2,4,4,512,409,1,1,0"),
    ("layer4 conv3.temporal 409->2048 +res", "8,2,4,4,409,2048,1,1,0,res"),
    ("cfg3 layer3 conv2.spatial (1,3,3) 256->576", "8,4,7,7,256,576,133,1,1"),
    ("cfg3 layer3 conv2.temporal (3,1,1) 576->256", "8,4,7,7,576,256,311,1,1"),
    ("cfg2 layer3 conv1 1024->256 (M=3136)", "8,2,14,14,1024,256,1,1,0"),
    ("cfg2 layer3 conv3 256->1024 +res", "8,2,14,14,256,1024,1,1,0,res"),
    ("cfg2 layer3 conv2 3x3x3 256->256", "8,2,14,14,256,256,333,1,1"),
    ("cfg2 layer4 conv1 2048->512 (M=392)", "8,1,7,7,2048,512,1,1,0"),
]


def main():
    import io
    import contextlib
    n = lib.ptx_conv3d_num_configs()
    cfgs = []
    for c in range(n):
        name = lib.ptx_conv3d_config_name(c).decode()
        if "f16" in name or "x3" in name or "direct" in name or "kwr" in name or "chain" in name:
            continue
        cfgs.append(c)
    for title, spec in PROBLEMS:
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            cm.run(spec, cfgs, [1, 2, 4, 8], iters=20)
        rows = []
        head = ""
        for line in buf.getvalue().splitlines():
            if line.startswith("##"):
                head = line
            elif " ms " in line:
                f = line.split()
                rows.append((float(f[f.index("ms") - 1]), line))
        rows.sort()
        print("== %s   %s" % (title, head))
        for _, line in rows[:6]:
            print(line)
        sys.stdout.flush()


if __name__ == "__main__":
    main()
