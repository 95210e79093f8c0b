"""The reference's own known-answer tests, transcribed into oracle/selftest.cpp, must pass against the oracle.
They are what pins the CPU restatement: the Rust reference cannot run here."""
import ctypes

import pytest

from oracle import binding

NAMES = [binding.lib().oracle_selftest_name(i).decode() for i in range(binding.lib().oracle_selftest_count())]


def test_every_reference_kat_is_transcribed():
    expected = {"local_trigonometry_test", "fresnel_test", "specular_refl_test", "diffuse_refl_test", "play_with_mf_brdf",
                "diff_area_validate", "pdf_integral_validate", "beckmann_rho", "quad_frame_test", "custom_frame_test", "sphere_test",
                "tricky_triangle", "reflect_refract_test", "float_doctests", "bbox_transform_test", "sphere_sample_pdf_integrate",
                "observe_sphere_sample_towards", "lambertian_test", "mf_refl_test", "find_interval_test", "catmull_test",
                "fourier_sum_test"}
    assert expected <= set(NAMES)


@pytest.mark.parametrize("name", NAMES)
def test_reference_kat(name):
    log = ctypes.create_string_buffer(8192)
    failures = binding.lib().oracle_selftest(name.encode(), log, 8192)
    assert failures == 0, log.value.decode()
