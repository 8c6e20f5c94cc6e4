"""CPU: config / registry semantics the reference's tests pin (test/torch/test_config.py bodies, test/common/test_common.py)."""

import torch

from neural_compressor_amd.common import ComposableConfig, config_registry
from neural_compressor_amd.torch.quantization import AWQConfig, GPTQConfig, RTNConfig
from neural_compressor_amd.torch.utils.utility import algos_mapping


def build_simple_torch_model():
    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1 = torch.nn.Linear(30, 50)
            self.fc2 = torch.nn.Linear(50, 30)
            self.fc3 = torch.nn.Linear(30, 5)

        def forward(self, x):
            return self.fc3(self.fc2(self.fc1(x)))

    return Model()


def test_defaults_match_reference():
    r, g, a = RTNConfig(), GPTQConfig(), AWQConfig()
    assert (r.dtype, r.bits, r.use_sym, r.group_size, r.group_dim, r.use_layer_wise) == ("int", 4, True, 32, 1, True)
    assert (g.bits, g.group_size, g.block_size, g.percdamp, g.act_order, g.use_layer_wise) == (4, 32, 2048, 0.01, False, False)
    assert (a.use_auto_scale, a.use_auto_clip, a.folding, a.double_quant_use_sym) == (True, True, False, True)
    assert set(algos_mapping) >= {"rtn", "gptq", "awq"}
    assert set(config_registry.get_cls_configs()["torch"]) >= {"rtn", "gptq", "awq"}


def test_config_white_lst2():
    model = build_simple_torch_model()
    global_config = RTNConfig(bits=4, dtype="nf4")
    fc1_config = RTNConfig(bits=6, dtype="int8", white_list=["fc1"])
    quant_config = global_config + fc1_config
    mapping = quant_config.to_config_mapping(model_info=quant_config.get_model_info(model))
    assert mapping[("fc1", "Linear")].bits == 6
    assert mapping[("fc2", "Linear")].bits == 4


def test_config_from_dict_and_to_dict():
    quant_config = {"rtn": {"global": {"dtype": "nf4", "bits": 4, "group_size": 32}, "local": {"fc1": {"dtype": "int8", "bits": 4}}}}
    cfg = RTNConfig.from_dict(quant_config["rtn"])
    assert cfg.local_config is not None and cfg.local_config["fc1"].dtype == "int8"
    d = RTNConfig(dtype="nf4", bits=4, group_size=32).to_dict()
    assert d["bits"] == 4 and "params_list" not in d
    assert RTNConfig.from_dict(d).group_size == 32


def test_same_type_configs_addition():
    q1 = RTNConfig.from_dict({"dtype": "nf4", "bits": 4, "group_size": 32})
    q2 = RTNConfig.from_dict({"global": {"bits": 8, "group_size": 32}, "local": {"fc1": {"dtype": "int8", "bits": 4}}})
    q = q1 + q2
    d = q.to_dict()
    assert d["global"]["bits"] == 4 and d["local"]["fc1"]["dtype"] == "int8"


def test_diff_types_configs_addition():
    q = RTNConfig(bits=8) + GPTQConfig(bits=4)
    assert isinstance(q, ComposableConfig)
    d = q.to_dict()
    assert d["rtn"]["bits"] == 8 and d["gptq"]["bits"] == 4


def test_config_mapping_and_set_local():
    model = build_simple_torch_model()
    cfg = RTNConfig(bits=4, dtype="nf4")
    cfg.set_local("fc1", RTNConfig(bits=6, dtype="int8"))
    m = cfg.to_config_mapping(model_info=cfg.get_model_info(model))
    assert m[("fc1", "Linear")].bits == 6 and m[("fc2", "Linear")].bits == 4
    cfg.set_local("fc2", RTNConfig(bits=3, dtype="int8"))
    m = cfg.to_config_mapping(model_info=cfg.get_model_info(model))
    assert m[("fc2", "Linear")].bits == 3 and m[("fc3", "Linear")].bits == 4
    cfg2 = RTNConfig(bits=4)
    cfg2.set_local(torch.nn.Linear, RTNConfig(bits=6))
    m = cfg2.to_config_mapping(model_info=cfg2.get_model_info(model))
    assert all(v.bits == 6 for v in m.values())


def test_lm_head_is_left_in_fp32_unless_requested():
    class LM(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.body = torch.nn.Linear(8, 8)
            self.lm_head = torch.nn.Linear(8, 16)

    m = LM()
    cfg = GPTQConfig()
    mp = cfg.to_config_mapping(model_info=cfg.get_model_info(m))
    assert mp[("lm_head", "Linear")].dtype == "fp32" and mp[("body", "Linear")].dtype == "int"
    cfg = RTNConfig(quant_lm_head=True)
    mp = cfg.to_config_mapping(model_info=cfg.get_model_info(m))
    assert mp[("lm_head", "Linear")].dtype == "int"


def test_expand_and_eq():
    cfgs = RTNConfig(bits=[4, 8], group_size=[32, 128]).expand()
    assert len(cfgs) == 4 and {c.bits for c in cfgs} == {4, 8}
    assert RTNConfig(bits=4) == RTNConfig(bits=4) and RTNConfig(bits=4) != RTNConfig(bits=8)
    g = GPTQConfig(act_order=True, percdamp=0.1, block_size=128, use_mse_search=False)
    assert g.to_dict()["percdamp"] == 0.1 and g.act_order


def test_2x_named_shim_translates_to_3x_configs():
    """`PostTrainingQuantConfig(approach="weight_only", op_type_dict=..., recipes=...)` (the 2.x vocabulary BASELINE.json's
    north_star uses) maps onto the 3.x config objects; `quantization.fit` is importable without a GPU."""
    from neural_compressor_amd import quantization
    from neural_compressor_amd.config import PostTrainingQuantConfig
    from neural_compressor_amd.torch.quantization import AWQConfig, GPTQConfig, RTNConfig

    c = PostTrainingQuantConfig(
        approach="weight_only",
        op_type_dict={".*": {"weight": {"bits": 4, "group_size": 128, "scheme": "sym", "algorithm": "GPTQ"}}},
        op_name_dict={".*lm_head": {"weight": {"dtype": "fp32"}}},
        recipes={"gptq_args": {"percdamp": 0.02, "block_size": 128, "act_order": True}},
    ).to_3x()
    assert isinstance(c, GPTQConfig) and (c.bits, c.group_size, c.use_sym, c.percdamp, c.act_order) == (4, 128, True, 0.02, True)
    assert ".*lm_head" in c.local_config and c.local_config[".*lm_head"].dtype == "fp32"
    r = PostTrainingQuantConfig(op_type_dict={".*": {"weight": {"bits": 8, "group_size": -1, "scheme": "asym", "algorithm": "RTN"}}},
                                recipes={"rtn_args": {"enable_mse_search": True}}).to_3x()
    assert isinstance(r, RTNConfig) and (r.bits, r.group_size, r.use_sym, r.use_mse_search) == (8, -1, False, True)
    a = PostTrainingQuantConfig(op_type_dict={".*": {"weight": {"bits": 4, "group_size": 32, "scheme": "asym", "algorithm": "AWQ"}}},
                                recipes={"awq_args": {"enable_auto_scale": False, "folding": True}}).to_3x()
    assert isinstance(a, AWQConfig) and (a.use_auto_scale, a.use_auto_clip, a.folding) == (False, True, True)
    import pytest

    with pytest.raises(NotImplementedError):
        PostTrainingQuantConfig(approach="static")
    with pytest.raises(ValueError):
        quantization.fit(None, PostTrainingQuantConfig(op_type_dict={".*": {"weight": {"algorithm": "GPTQ"}}}))
