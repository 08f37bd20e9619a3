"""Host side of the sampler: time pairs and noise-schedule scalars.

These are a handful of scalars per step, evaluated with torch CPU fp32 ops in exactly the
reference's op order because ``alpha_cosine_log_snr(1.0)`` is ill-conditioned (fp32 gives
-18.9075, fp64 -18.9043; the difference moves the final logits by 5e-3 - SURVEY.md §7 hard
part 8).  The scalars are handed to the HIP library in ``ddp_step`` records
(include/ddp_mi355x.h); everything downstream of them runs on the GPU.

Reference: segmentation/mmseg/models/segmentors/ddp.py:14-28 (schedules), :204-213 (time pairs),
:225-231 (alpha/sigma), :276-283 (ddpm); depth/depth/models/depther/ddp.py:207-227.
"""
import math

import torch
from torch.special import expm1


def log(t, eps=1e-20):
    return torch.log(t.clamp(min=eps))


def beta_linear_log_snr(t):
    return -torch.log(expm1(1e-4 + 10 * (t ** 2)))


def alpha_cosine_log_snr(t, ns=0.0002, ds=0.00025):
    return -log((torch.cos((t + ns) / (1 + ds) * math.pi * 0.5) ** -2) - 1, eps=1e-5)


def log_snr_to_alpha_sigma(log_snr):
    return torch.sqrt(torch.sigmoid(log_snr)), torch.sqrt(torch.sigmoid(-log_snr))


def gamma(t, ns=0.0002, ds=0.00025):
    return torch.cos(((t + ns) / (1 + ds)) * math.pi / 2) ** 2


NOISE_SCHEDULES = {'linear': beta_linear_log_snr, 'cosine': alpha_cosine_log_snr}


def get_sampling_timesteps(timesteps, time_difference=1, sample_range0=0.0):
    """[(t_now, t_next)] as python floats (ddp.py:204-213)."""
    times = []
    for step in range(timesteps):
        t_now = 1 - (step / timesteps) * (1 - sample_range0)
        t_next = max(1 - (step + 1 + time_difference) / timesteps * (1 - sample_range0), sample_range0)
        times.append((t_now, t_next))
    return times


def step_records(task, timesteps, time_difference=1, sample_range0=0.0, noise_schedule='cosine', sampler='ddim'):
    """-> list of dicts with the fields of ``ddp_step``."""
    if noise_schedule not in NOISE_SCHEDULES:
        raise ValueError(f'invalid noise schedule {noise_schedule}')
    log_snr_fn = NOISE_SCHEDULES[noise_schedule]
    recs = []
    for t_now, t_next in get_sampling_timesteps(timesteps, time_difference, 0.0 if task != 'seg' else sample_range0):
        tn = torch.tensor([t_now], dtype=torch.float32)
        tx = torch.tensor([t_next], dtype=torch.float32)
        r = dict(time_in=0.0, alpha=0.0, sigma=0.0, alpha_next=0.0, sigma_next=0.0, ddpm_c=0.0, ddpm_std=0.0,
                 ddpm_add_noise=0)
        if task == 'depth':
            a_now, a_next = gamma(tn), gamma(tx)
            r.update(time_in=float(tn), alpha=float(a_now.sqrt()), sigma=float(1 / (1 - a_now).sqrt()),
                     alpha_next=float(a_next.sqrt()), sigma_next=float((1 - a_next).sqrt()))
        else:
            ls, lsn = log_snr_fn(tn), log_snr_fn(tx)
            alpha, sigma = log_snr_to_alpha_sigma(ls)
            alpha_next, sigma_next = log_snr_to_alpha_sigma(lsn)
            r.update(time_in=float(ls), alpha=float(alpha), sigma=float(sigma), alpha_next=float(alpha_next),
                     sigma_next=float(sigma_next))
            if sampler == 'ddpm':
                c = -expm1(ls - lsn)
                variance = (sigma_next ** 2) * c
                r.update(ddpm_c=float(c), ddpm_std=float((0.5 * log(variance)).exp()),
                         ddpm_add_noise=int(t_next > 0))
        recs.append(r)
    return recs
