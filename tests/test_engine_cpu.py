"""Host logic of the training-runtime counterparts (CPU): schedules, parameter groups / layer decay,
grad-norm definition, scaler interface, step ordering of train_one_epoch."""
import numpy as np
import torch
import torch.nn as nn

import golden_recipe as R
import simple_tad_amd as T
from simple_tad_amd import engine as E


def test_cosine_scheduler_known_answer(golden):
    g = golden("g5_schedules")
    assert np.array_equal(E.cosine_scheduler(1e-3, 1e-6, 3, 10, warmup_epochs=1), g["cos_1e-3_1e-6_3_10_1"])
    assert np.array_equal(E.cosine_scheduler(5e-4, 1e-6, 2, 7, warmup_epochs=0), g["cos_5e-4_1e-6_2_7_0"])
    assert np.array_equal(E.cosine_scheduler(1e-3, 1e-5, 4, 5, warmup_epochs=1, start_warmup_value=1e-6, warmup_steps=3),
                          g["cos_warmup_steps"])


def test_layer_decay_and_param_groups(golden):
    g = golden("g5_schedules")
    names = [str(n) for n in g["layer_names"]]
    assert [E.get_num_layer_for_vit(n, 14) for n in names] == list(g["layer_ids"])
    a = E.LayerDecayValueAssigner.from_decay(0.75, 12)
    assert np.array_equal(np.array(a.values), g["layer_scales_0.75_12"])
    m = T.VisionTransformer(img_size=16, patch_size=8, embed_dim=128, depth=2, num_heads=2, qkv_bias=True, all_frames=4, num_classes=2)
    a2 = E.LayerDecayValueAssigner.from_decay(0.6, m.get_num_layers())
    groups = E.get_parameter_groups(m, 0.05, m.no_weight_decay(), a2.get_layer_id, a2.get_scale)
    # layers 0 (patch_embed), 1, 2 (blocks), 3 (fc_norm/head) x {decay, no_decay}
    assert len(groups) == 8
    by_scale = sorted({round(gr["lr_scale"], 6) for gr in groups})
    assert by_scale == sorted({round(0.6 ** (3 - i), 6) for i in range(4)})
    n_params = sum(len(gr["params"]) for gr in groups)
    assert n_params == len(list(m.parameters()))
    for gr in groups:
        for p in gr["params"]:
            assert (p.dim() == 1) == (gr["weight_decay"] == 0.0)
    opt = E.create_optimizer(m, lr=1e-3, weight_decay=0.05, layer_decay=0.6)
    assert len(opt.param_groups) == 8 and all("lr_scale" in gr for gr in opt.param_groups)


def test_grad_norm_definition(golden):
    g = golden("g5_schedules")
    ps = [nn.Parameter(torch.zeros(3, 4)), nn.Parameter(torch.zeros(5))]
    ps[0].grad = torch.full((3, 4), 0.5)
    ps[1].grad = torch.arange(5, dtype=torch.float32)
    assert abs(E.get_grad_norm_(ps).item() - float(g["grad_norm_known"])) < 1e-6


def test_train_one_epoch_step_ordering():
    """drives the loop with a plain torch model on CPU: lr/wd per step from the tables, update_freq accumulation"""
    torch.manual_seed(0)
    model = nn.Sequential(nn.Flatten(), nn.Linear(3 * 2 * 4 * 4, 2))
    opt = torch.optim.AdamW([{"params": list(model.parameters()), "lr_scale": 0.5, "weight_decay": 0.05}], lr=1.0)
    lr_sched = E.cosine_scheduler(1e-2, 1e-4, 1, 3, warmup_epochs=0)
    wd_sched = E.cosine_scheduler(0.05, 0.01, 1, 3, warmup_epochs=0)
    data = [(torch.randn(4, 3, 2, 4, 4), torch.randint(0, 2, (4,)), None, None) for _ in range(6)]
    seen = []
    stats = E.train_one_epoch(model, nn.CrossEntropyLoss(), data, opt, torch.device("cpu"), 0, E.NativeScalerWithGradNormCount(),
                              lr_schedule_values=lr_sched, wd_schedule_values=wd_sched, num_training_steps_per_epoch=3,
                              update_freq=2, log=lambda e, i, s: seen.append((i, opt.param_groups[0]["lr"], opt.param_groups[0]["weight_decay"])))
    assert len(stats["loss"]) == 6
    assert [s is None for s in stats["grad_norm"]] == [True, False] * 3       # norm only on update steps
    for i, lr, wd in seen:
        assert abs(lr - lr_sched[i // 2] * 0.5) < 1e-12 and abs(wd - wd_sched[i // 2]) < 1e-12
    assert all(s == 1.0 for s in stats["loss_scale"])
    assert all(p.grad is not None and float(p.grad.abs().sum()) == 0.0 for p in model.parameters())  # zero_grad after update


def test_loss_scaler_state_round_trips_across_precision_modes():
    """the carried loss scale survives a checkpoint whatever the precision mode was on either side; the logged ``scale``
    stays GradScaler's key and reads 1.0 while no scaling is applied (engine_for_finetuning.py:100 logs it every step)"""
    import warnings
    from simple_tad_amd import engine as E
    half = E.NativeScalerWithGradNormCount(enabled=True)
    half.scale, half.growth_tracker = 2.0 ** 20, 7
    sd = half.state_dict()
    assert sd["scale"] == 2.0 ** 20 and sd["_scale"] == 2.0 ** 20 and sd["_growth_tracker"] == 7
    # restored into a run whose precision is (still) not half: the scale must not be dropped ...
    later = E.NativeScalerWithGradNormCount(enabled=False)
    later.load_state_dict(sd)
    assert later.scale == 2.0 ** 20 and later.growth_tracker == 7 and later.state_dict()["scale"] == 1.0
    later.enabled = True  # ... so that switching to half afterwards continues from it
    assert later.state_dict()["scale"] == 2.0 ** 20
    # a checkpoint written without loss scaling carries the 1.0 placeholder: it must not install scale 1.0 in a half-mode run
    plain = E.NativeScalerWithGradNormCount(enabled=False).state_dict()
    assert plain["scale"] == 1.0 and plain["_scale"] == 65536.0
    resumed = E.NativeScalerWithGradNormCount(enabled=True)
    resumed.load_state_dict(plain)
    assert resumed.scale == 65536.0
    with warnings.catch_warnings(record=True) as w:  # the reference's GradScaler.state_dict() of an unscaled run (no `_scale`)
        warnings.simplefilter("always")
        legacy = E.NativeScalerWithGradNormCount(enabled=True)
        legacy.load_state_dict({"scale": 1.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000, "_growth_tracker": 3})
        assert legacy.scale == 65536.0 and legacy.growth_tracker == 3 and any("loss scale 1.0" in str(x.message) for x in w)
    legacy.load_state_dict({"scale": 4096.0})
    assert legacy.scale == 4096.0


def test_loss_scaling_with_a_plain_optimizer_needs_the_parameter_list():
    import pytest
    import torch
    from simple_tad_amd import engine as E
    lin = torch.nn.Linear(4, 2)
    opt = torch.optim.SGD(lin.parameters(), lr=0.1)
    sc = E.NativeScalerWithGradNormCount(enabled=True)
    with pytest.raises(ValueError, match="parameters"):
        sc(lin(torch.randn(3, 4)).sum(), opt, parameters=None)
    w0 = lin.weight.detach().clone()
    sc(lin(torch.randn(3, 4)).sum(), opt, parameters=list(lin.parameters()))
    assert not torch.equal(w0, lin.weight) and sc.scale == 65536.0 and sc.growth_tracker == 1
    # an overflowing step is skipped and halves the scale
    sc2 = E.NativeScalerWithGradNormCount(enabled=True)
    lin2 = torch.nn.Linear(4, 2)
    opt2 = torch.optim.SGD(lin2.parameters(), lr=0.1)
    before = lin2.weight.detach().clone()
    n = sc2((lin2(torch.ones(3, 4)) * float("inf")).sum(), opt2, parameters=list(lin2.parameters()))
    assert not torch.isfinite(n) and sc2.scale == 32768.0 and sc2.skipped_steps == 1 and torch.equal(before, lin2.weight)
