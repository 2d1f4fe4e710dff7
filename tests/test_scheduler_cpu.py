"""CPU: product scheduler (host logic) vs the oracle scheduler and the Appendix-E table."""
import torch

from oracle.scheduler import EulerDiscreteScheduler as OracleSched
from this_and_that_vdm_amd.svd.scheduling_euler_discrete import EulerDiscreteScheduler


def test_schedule_tables_identical():
    a, b = EulerDiscreteScheduler(), OracleSched()
    a.set_timesteps(25)
    b.set_timesteps(25)
    torch.testing.assert_close(a.sigmas, b.sigmas, rtol=0, atol=0)
    torch.testing.assert_close(a.timesteps, b.timesteps, rtol=1e-6, atol=1e-7)
    assert abs(float(a.init_noise_sigma) - 700.000732) < 1e-3


def test_step_matches_oracle():
    a, b = EulerDiscreteScheduler(), OracleSched()
    a.set_timesteps(7)
    b.set_timesteps(7)
    x = torch.randn(1, 3, 4, 5, 5, generator=torch.Generator().manual_seed(0)) * 700
    for t in a.timesteps:
        v = torch.randn(x.shape, generator=torch.Generator().manual_seed(1))
        torch.testing.assert_close(a.scale_model_input(x, t), b.scale_model_input(x, t))
        xa, xb = a.step(v, t, x).prev_sample, b.step(v, t, x)
        torch.testing.assert_close(xa, xb, rtol=1e-6, atol=1e-6)
        x = xa
