"""Build-container check that reference environments drop in UNCHANGED (north star; SURVEY 8b 'env-side boundary'):
the reference's own env/env_mock.hpp -- which pulls the reference's env/env.hpp, common/serializable.hpp and json.hpp --
is compiled together with THIS repository's pooled VecEnv (host/env/vec_env.hpp) and Mat (host/mat.hpp through the
Eigen/Dense shim; the image has no Eigen), and the reference's VecEnv test (test/vecenv_test.cpp: simulate_steps + its
TEST_CASE for 1, 2 and 16 environments, under the reference's own Catch header) runs against it.  Nothing of the reference
is copied into the repository: the test lines are spliced from the mounted tree at test time; skipped elsewhere."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
HOST = os.path.join(ROOT, "ppo_cpp_amd", "host")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only mounted in the build container")


def _compile_and_run(tmp_path, src, name, extra=()):
    cpp = tmp_path / (name + ".cpp")
    cpp.write_text(src)
    exe = tmp_path / name
    cmd = ["g++", "-std=c++17", "-O1", "-pthread", "-I", os.path.join(HOST, "shim"), "-I", HOST, "-I", os.path.join(ROOT, "include"), *extra,
           "-o", str(exe), str(cpp)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    return subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)


def test_reference_env_mock_and_vecenv_test_run_against_the_kept_headers(tmp_path):
    lines = open(os.path.join(REF, "test", "vecenv_test.cpp")).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("typedef Eigen::Matrix"))
    body = "\n".join(lines[start:])                          # typedef Mat, simulate_steps, TEST_CASE
    src = """
#define CATCH_CONFIG_MAIN
#include <chrono>
#include <iostream>
#include <memory>
#include <thread>
#include "%s/test/catch.hpp"
#include "%s/env/env_mock.hpp"        // the reference's file, unchanged (brings the reference's env.hpp / serializable.hpp / json.hpp)
#include "env/vec_env.hpp"            // this repository's pooled VecEnv behind the same class interface
%s
""" % (REF, REF, body)
    # (the reference's 2019 Catch header predates glibc's non-constant MINSIGSTKSZ: its signal handler is switched off)
    r = _compile_and_run(tmp_path, src, "vecenv_ref_test", extra=("-DCATCH_CONFIG_NO_POSIX_SIGNALS",))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "All tests passed" in r.stdout


def test_mat_shim_covers_the_eigen_surface_the_reference_headers_use(tmp_path):
    """The expressions of env_normalize.hpp:64-116 and running_statistics.hpp:26-104, written exactly as the reference writes
    them, evaluate to the closed-form answers on the stand-in Mat."""
    src = r"""
#include <Eigen/Dense>
#include <cassert>
#include <cmath>
#include <cstdio>
typedef Eigen::Matrix<float, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> Mat;
static bool close(float a, float b) { return std::fabs(a - b) <= 1e-5f * (1.f + std::fabs(b)); }
int main() {
    Mat x(3, 2); x(0,0)=1; x(0,1)=2; x(1,0)=3; x(1,1)=4; x(2,0)=5; x(2,1)=9;
    Mat mean = x.colwise().mean();                                            // running_statistics.hpp:38
    assert(mean.rows() == 1 && close(mean(0,0), 3.f) && close(mean(0,1), 5.f));
    Mat dev = x.rowwise() - mean.row(0);                                      // :45, env_normalize.hpp:100
    Mat var = dev.cwiseProduct(dev).colwise().sum() * (1.f / 3.f);
    assert(close(var(0,0), 8.f/3.f) && close(var(0,1), 26.f/3.f));
    const float eps = 1e-8f;
    Mat o = (x.rowwise() - mean.row(0)) * (var.row(0) + eps * Mat::Ones(1, x.cols())).cwiseSqrt().cwiseInverse().row(0).asDiagonal();
    assert(close(o(2,1), (9.f - 5.f) / std::sqrt(26.f/3.f)));
    Mat rews = Mat::Ones(3, 1) * 2.f;
    rews *= (var.block(0,0,1,1) + eps * Mat::Ones(1, 1)).cwiseSqrt().cwiseInverse().row(0).asDiagonal();   // env_normalize.hpp:79
    assert(close(rews(1,0), 2.f / std::sqrt(8.f/3.f)));
    Mat ret = Mat::Zero(3,1); ret = ret * 0.99f + rews;                       // :66
    ret = ret.cwiseProduct(Mat::Ones(3,1) - Mat::Constant(3,1,1.f));          // :91
    assert(ret.squaredNorm() == 0.f);
    Mat clipped = x.cwiseMin(4.f).cwiseMax(2.f);                              // matrix_clamp.hpp:32-35
    assert(clipped(0,0) == 2.f && clipped(2,1) == 4.f);
    x.row(0) = mean; x.col(1) = Mat::Zero(3,1);                               // l-value views (vec_env.hpp:102,118)
    assert(close(x(0,0), 3.f) && x(2,1) == 0.f);
    Mat t = x.transpose(); assert(t.rows() == 2 && t(0,2) == 5.f);
    t.transposeInPlace(); assert(t.rows() == 3);
    float buf[4] = {1,2,3,4};
    Mat m = Eigen::Map<Mat>(buf, 2, 2);                                       // running_statistics.hpp:77-81
    assert(m(1,0) == 3.f && close((m * m)(0,0), 7.f) && close((-m).sum(), -10.f));
    auto expr = 2.0 * Mat::Ones(2, 3);                                        // env_mock.hpp:44,50
    assert(expr.rows() == 2 && expr(1,2) == 2.f);
    std::puts("ok");
    return 0;
}
"""
    r = _compile_and_run(tmp_path, src, "mat_surface")
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
