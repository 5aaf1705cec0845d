"""CPU test of the GPU BVH builder and of the traversal algorithm the kernels run (restated on the
host in tests/cpp/test_bvh.cpp): structure validation, closest hit == list traversal for random
scenes and rays (incl. one-primitive trees, coincident centres, zero direction components), and the
stack / candidate-queue bounds the kernel's LDS arrays rely on."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bvh_builder_and_traversal(tmp_path):
    exe = str(tmp_path / "test_bvh")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-ffp-contract=off", os.path.join(ROOT, "tests", "cpp", "test_bvh.cpp"),
                           "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok:"), out.stdout


def test_reciprocal_division_equals_ieee(tmp_path):
    """div_recip (rt_device.h) on the host: random operands and the camera / sphere-normal shapes."""
    exe = str(tmp_path / "test_div_recip")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-ffp-contract=off", os.path.join(ROOT, "tests", "cpp", "test_div_recip.cpp"),
                           "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok:"), out.stdout
