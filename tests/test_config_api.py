"""CPU: config / registry semantics the reference's tests pin (test/torch/test_config.py bodies, test/common/test_common.py)."""

import pytest
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


def test_front_end_configs_roundtrip(tmp_path):
    """neural_compressor.transformers config classes: defaults, serialised keys, reload (quantization_config.py:242-456)."""
    from neural_compressor_amd.transformers import AwqConfig, GPTQConfig, RtnConfig, TeqConfig
    from neural_compressor_amd.transformers.utils import QUANT_CONFIG

    r = RtnConfig()
    assert (r.bits, r.group_size, r.sym, r.scheme, r.weight_dtype) == (4, 32, True, "sym", "int4")
    assert r.modules_to_not_convert == ["lm_head", "transformer.output_layer", "embed_out"]
    assert RtnConfig(quant_lm_head=True).modules_to_not_convert == []
    g = GPTQConfig(bits=8, group_size=128, desc_act=True, damp_percent=0.01, tokenizer=object(), dataset=[1, 2])
    assert g.weight_dtype == "int8" and g.to_diff_dict() == {"bits": 8, "weight_dtype": "int8", "group_size": 128, "desc_act": True, "damp_percent": 0.01}
    with pytest.raises(ValueError):
        GPTQConfig(damp_percent=1.5)
    with pytest.raises(ValueError):
        GPTQConfig(bits=3)
    a = AwqConfig(zero_point=False)
    assert a.sym is True and a.scheme == "sym" and AwqConfig().sym is False
    assert TeqConfig().quant_method.value == "teq"
    g.post_init()
    assert g.compute_dtype == "fp16" and g.scale_dtype == "fp16"
    g.remove_redundant_parameters()
    assert not hasattr(g, "tokenizer") and not hasattr(g, "dataset") and not hasattr(g, "static_groups")
    g.save_pretrained(str(tmp_path))
    assert QUANT_CONFIG == "quantize_config.json" and (tmp_path / QUANT_CONFIG).exists()
    g2 = GPTQConfig.from_pretrained(str(tmp_path))
    assert g2.bits == 8 and g2.group_size == 128 and g2.desc_act is True and g2.quant_method.value == "gptq"
    extra = AwqConfig.from_dict({"quant_method": "awq", "bits": 4, "group_size": 128, "zero_point": True, "version": "gemm"})
    assert extra.version == "gemm" and extra.group_size == 128
    unused = r.update(bits=8, not_a_field=1)
    assert r.bits == 8 and unused == {"not_a_field": 1}


def test_2x_shim_maps_the_smooth_quant_recipe():
    from neural_compressor_amd.config import PostTrainingQuantConfig

    conf = PostTrainingQuantConfig(approach="static", recipes={"smooth_quant": True, "smooth_quant_args": {"alpha": 0.7, "folding": True}},
                                   op_name_dict={"lm_head": {"weight": {"dtype": "fp32"}, "activation": {"dtype": "fp32"}}})
    cfg = conf.to_3x()
    assert cfg.name == "smooth_quant" and cfg.alpha == 0.7 and cfg.folding is True
    model = build_simple_torch_model()
    mapping = cfg.to_config_mapping(model_info=cfg.get_model_info(model))
    assert all(c.name == "smooth_quant" for c in mapping.values())
    with pytest.raises(NotImplementedError):
        PostTrainingQuantConfig(approach="static")  # plain static INT8 is out of scope


def test_short_rows_are_dropped_like_the_reference():
    from neural_compressor_amd.transformers.quantization.utils import _as_batches

    rows = [torch.arange(40), torch.arange(10), torch.arange(64).reshape(2, 32)]
    got = list(_as_batches(rows, None, 32, 100, 2))
    assert [tuple(b.shape) for b in got] == [(2, 32), (1, 32)]
    with pytest.raises(AssertionError):
        list(_as_batches([torch.arange(4)], None, 32, 8, 2))


def test_gptq_forward_groups_stack_only_compatible_batches(monkeypatch):
    """RAWGPTQuantizer._forward_groups (pure host logic): cached batches are stacked only when everything but the hidden
    states is the same tensor value, the leading dimension is 1 and the shapes agree; INC_MI355X_GPTQ_FORWARD_BATCH bounds it."""
    from neural_compressor_amd.torch.algorithms.weight_only.gptq import RAWGPTQuantizer

    rq = object.__new__(RAWGPTQuantizer)
    pos = torch.arange(8).view(1, 8)
    hs = [torch.randn(1, 8, 4) for _ in range(6)]
    hs[4] = torch.randn(1, 5, 4)  # a shorter sample cannot ride with the others
    rq.cache_positional_arguments = [list(hs)]
    rq.cache_key_arguments = {"position_ids": [pos.clone() for _ in range(6)], "mask": [None] * 6}
    assert rq._forward_groups(6, in_kwargs=False) == [[0, 1, 2, 3], [4], [5]]
    assert rq._forward_groups(6, in_kwargs=False) == [[0, 1, 2, 3], [4], [5]]  # cached
    rq2 = object.__new__(RAWGPTQuantizer)
    rq2.cache_positional_arguments = []
    rq2.cache_key_arguments = {"hidden_states": [torch.randn(1, 8, 4) for _ in range(5)],
                               "position_ids": [pos, pos, pos + 1, pos + 1, pos + 1]}  # different positions split the run
    monkeypatch.setenv("INC_MI355X_GPTQ_FORWARD_BATCH", "8")
    assert rq2._forward_groups(5, in_kwargs=True) == [[0, 1], [2, 3, 4]]
    rq3 = object.__new__(RAWGPTQuantizer)
    rq3.cache_positional_arguments = [[torch.randn(2, 8, 4) for _ in range(3)]]  # user batches of 2: never stacked
    rq3.cache_key_arguments = {}
    assert rq3._forward_groups(3, in_kwargs=False) == [[0], [1], [2]]
    monkeypatch.setenv("INC_MI355X_GPTQ_FORWARD_BATCH", "1")
    rq4 = object.__new__(RAWGPTQuantizer)
    rq4.cache_positional_arguments = [[torch.randn(1, 8, 4) for _ in range(3)]]
    rq4.cache_key_arguments = {}
    assert rq4._forward_groups(3, in_kwargs=False) == [[0], [1], [2]]


def test_awq_search_stacking_helpers(monkeypatch):
    from neural_compressor_amd.torch.algorithms.weight_only.awq import ActAwareWeightQuant

    aw = object.__new__(ActAwareWeightQuant)
    monkeypatch.setenv("INC_MI355X_AWQ_SEARCH_BATCH", "4")
    xs = [torch.randn(1, 3, 2) for _ in range(6)]
    st = aw._stack(xs)
    assert [(tuple(t.shape), n) for t, n in st] == [((4, 3, 2), 4), ((2, 3, 2), 2)]
    assert torch.equal(st[1][0], torch.cat(xs[4:], 0))
    assert [n for _, n in aw._stack(xs[:5] + [torch.randn(1, 4, 2)])] == [1] * 6  # ragged: one at a time
    p = torch.arange(3)
    assert aw._same_kwargs({"a": p, "b": (p, None), "c": 1}, {"a": p.clone(), "b": (p.clone(), None), "c": 1})
    assert not aw._same_kwargs({"a": p}, {"a": p + 1})
    assert not aw._same_kwargs({"a": p}, {"b": p})


def test_hf_config_export_and_device_map_checks():
    from neural_compressor_amd.torch.algorithms.weight_only.save_load import change_config_to_hf_format
    from neural_compressor_amd.transformers.models.modeling_auto import _device_of

    g = GPTQConfig(bits=4, group_size=128, use_sym=True, act_order=True, percdamp=0.02, true_sequential=True)
    fp = GPTQConfig(dtype="fp32")
    hf = change_config_to_hf_format({("model.layers.0.q", "Linear"): g, ("model.layers.0.k", "Linear"): g, ("lm_head", "Linear"): fp})
    assert (hf["bits"], hf["group_size"], hf["sym"], hf["desc_act"], hf["damp_percent"], hf["true_sequential"], hf["quant_method"]) == (
        4, 128, True, True, 0.02, True, "gptq")
    with pytest.raises(ValueError):
        change_config_to_hf_format({("lm_head", "Linear"): g})  # a quantised lm_head cannot be expressed in that format
    with pytest.raises(AssertionError):
        change_config_to_hf_format({("a", "Linear"): g, ("b", "Linear"): GPTQConfig(bits=8, group_size=128)})
    assert str(_device_of({"": "cuda:0"})) == "cuda:0" and str(_device_of("auto")) == "cuda"
    with pytest.raises(RuntimeError):
        _device_of("cpu")


def test_accelerator_registry_selection_order(monkeypatch):
    """Reference auto_accelerator.py:100-168, 427-456: register_accelerator(name, priority), the Auto_Accelerator interface and
    the selection order INC_TARGET_DEVICE (case-insensitive) > device_name > priority; and what is deliberately different
    here: no CPU class exists, so forcing the CPU path raises instead of silently running something else."""
    from neural_compressor_amd.torch.utils import auto_accelerator as A

    saved = dict(A.accelerator_registry.registered_accelerators)
    try:
        @A.register_accelerator(name="fakehip", priority=500)
        class Fake(A.HIPAccelerator):
            @classmethod
            def is_available(cls):
                return True

            def name(self):
                return "fakehip"

        monkeypatch.setattr(A.HIPAccelerator, "is_available", classmethod(lambda cls: True))
        monkeypatch.delenv("INC_TARGET_DEVICE", raising=False)
        A._select.cache_clear()
        assert A.auto_detect_accelerator().name() == "fakehip"            # highest priority wins
        assert A.auto_detect_accelerator("cuda").name() == "cuda"         # an explicit device name beats the priority
        assert A.auto_detect_accelerator("cuda:0").name() == "cuda"
        monkeypatch.setenv("INC_TARGET_DEVICE", "CUDA")
        A._select.cache_clear()
        assert A.auto_detect_accelerator("fakehip").name() == "cuda"      # the environment variable beats both
        monkeypatch.setenv("INC_TARGET_DEVICE", "cpu")
        A._select.cache_clear()
        with pytest.raises(RuntimeError, match="no CPU path"):
            A.auto_detect_accelerator()
        monkeypatch.delenv("INC_TARGET_DEVICE")
        A._select.cache_clear()
        with pytest.raises(RuntimeError, match="no CPU implementation"):
            A.auto_detect_accelerator("cpu")
        acc = A.HIPAccelerator()
        for method in ("is_available", "name", "device_name", "set_device", "current_device", "current_device_name", "device",
                       "empty_cache", "synchronize", "get_inc_accelerator_type"):
            assert callable(getattr(acc, method))
        assert acc.device_name(3) == "cuda:3" and acc.device_name() == "cuda"
    finally:
        A.accelerator_registry.registered_accelerators.clear()
        A.accelerator_registry.registered_accelerators.update(saved)
        A._select.cache_clear()


def test_prepared_gemm_call_is_a_cache_not_state():
    """ops.WoqGemmCall (the decode path's prepared inc_woq_gemm call kept in the module's __dict__) must not travel with copies or
    pickles of the module."""
    import copy
    import pickle

    from neural_compressor_amd.ops import WoqGemmCall

    obj = WoqGemmCall.__new__(WoqGemmCall)
    assert copy.deepcopy({"_call": obj})["_call"] is None
    assert pickle.loads(pickle.dumps({"_call": obj}))["_call"] is None
