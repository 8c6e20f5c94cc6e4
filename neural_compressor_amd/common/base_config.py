"""Config base + registry: the minimal restatement of the reference's config plug-in surface.

Mirrors (behaviour, not code) neural_compressor/common/base_config.py of the reference:
  ConfigRegistry / register_config   :57-187
  BaseConfig                         :190-680  (white list, global/local config, set_local, to_dict/from_dict,
                                                 `+`, expand, to_config_mapping with global -> op-type -> regex
                                                 op-name precedence :586-617)
  ComposableConfig                   :684-830
so that user code written against RTNConfig / GPTQConfig / AWQConfig keeps working unchanged.
"""

import inspect
import itertools
import json
import re
from collections import OrderedDict

from .utils import DEFAULT_WHITE_LIST, EMPTY_WHITE_LIST, logger

GLOBAL, LOCAL = "global", "local"
_INTERNAL_FIELDS = ("_global_config", "_local_config", "_white_list", "_is_initialized")


class ConfigRegistry:
    """framework name -> algorithm name -> {"priority", "cls"} (singleton, like the reference's)."""

    registered_configs = {}
    _instance = None

    def __new__(cls):
        if cls._instance is None:
            cls._instance = super().__new__(cls)
        return cls._instance

    @classmethod
    def register_config_impl(cls, framework_name, algo_name, priority=0):
        def deco(config_cls):
            cls.registered_configs.setdefault(framework_name, {})[algo_name] = {"priority": priority, "cls": config_cls}
            return config_cls

        return deco

    @classmethod
    def get_all_configs(cls):
        return cls.registered_configs

    @classmethod
    def get_sorted_configs(cls):
        out = OrderedDict()
        for fwk, algos in sorted(cls.registered_configs.items()):
            out[fwk] = OrderedDict(sorted(algos.items(), key=lambda kv: kv[1]["priority"], reverse=True))
        return out

    @classmethod
    def get_cls_configs(cls):
        return {fwk: {name: d["cls"] for name, d in algos.items()} for fwk, algos in cls.registered_configs.items()}

    @classmethod
    def get_all_config_cls_by_fwk_name(cls, fwk_name):
        return [d["cls"] for d in cls.registered_configs.get(fwk_name, {}).values()]


config_registry = ConfigRegistry()


def register_config(framework_name, algo_name, priority=0):
    """Class decorator registering an algorithm config (reference base_config.py:171)."""
    return config_registry.register_config_impl(framework_name, algo_name, priority)


class BaseConfig:
    """Algorithm config: a flat set of parameters + per-operator overrides."""

    name = "base_config"
    params_list = []
    non_tunable_params = ["white_list"]
    _is_initialized = False

    def __init__(self, white_list=DEFAULT_WHITE_LIST):
        object.__setattr__(self, "_global_config", None)
        object.__setattr__(self, "_local_config", {})
        object.__setattr__(self, "_white_list", white_list)

    # -- initialisation of the global / white-listed local configs --------------------------------
    def _post_init(self):
        if self.white_list == DEFAULT_WHITE_LIST:
            self._global_config = self.__class__(**self.get_params_dict(), white_list=None)
        elif isinstance(self.white_list, list) and len(self.white_list) > 0:
            for key in self.white_list:
                self.set_local(key, self.__class__(**self.get_params_dict(), white_list=None))
        elif self.white_list == EMPTY_WHITE_LIST:
            return
        else:
            raise NotImplementedError(
                f"white_list must be {DEFAULT_WHITE_LIST!r}, {EMPTY_WHITE_LIST!r} or a non-empty list, got {self.white_list!r}"
            )
        object.__setattr__(self, "_is_initialized", True)

    def __setattr__(self, key, value):
        object.__setattr__(self, key, value)
        if self._is_initialized and key in self.params_list:
            # a tunable parameter changed after construction: rebuild the derived global config
            object.__setattr__(self, "_is_initialized", False)
            self._post_init()

    @classmethod
    def _generate_params_list(cls):
        names = list(inspect.signature(cls.__init__).parameters)[1:]
        return [n for n in names if n not in ("args", "kwargs")]

    # -- properties ---------------------------------------------------------------------------------
    @property
    def white_list(self):
        return self._white_list

    @white_list.setter
    def white_list(self, value):
        object.__setattr__(self, "_white_list", value)

    @property
    def global_config(self):
        return self._global_config

    @global_config.setter
    def global_config(self, cfg):
        object.__setattr__(self, "_global_config", cfg)

    @property
    def local_config(self):
        return self._local_config

    @local_config.setter
    def local_config(self, cfg):
        object.__setattr__(self, "_local_config", cfg)

    def set_local(self, operator_name_or_list, config):
        """Attach `config` to an operator name / regex / module type (or a list of them)."""
        keys = operator_name_or_list if isinstance(operator_name_or_list, list) else [operator_name_or_list]
        for key in keys:
            if key in self.local_config:
                logger.warning("The configuration for %s has already been set, update it.", key)
            self.local_config[key] = config
        return self

    # -- (de)serialisation --------------------------------------------------------------------------
    def get_params_dict(self):
        return {k: v for k, v in self.__dict__.items() if k not in _INTERNAL_FIELDS}

    def to_dict(self):
        params = self.get_params_dict()
        if self.local_config:
            out = {LOCAL: {str(k): v.to_dict() for k, v in self.local_config.items()}}
            if self.global_config:
                out[GLOBAL] = params
        else:
            out = params
        out.pop("params_list", None)
        return out

    @classmethod
    def from_dict(cls, config_dict):
        if GLOBAL not in config_dict and LOCAL not in config_dict:
            return cls(**config_dict)
        config = cls(**config_dict.get(GLOBAL, {}))
        for op_name, op_cfg in config_dict.get(LOCAL, {}).items():
            config.set_local(op_name, cls(**op_cfg))
        return config

    @classmethod
    def from_json_file(cls, filename):
        with open(filename, "r", encoding="utf-8") as f:
            return cls.from_dict(json.load(f))

    def to_json_file(self, filename):
        with open(filename, "w", encoding="utf-8") as f:
            json.dump(self.to_dict(), f, indent=4)

    def to_json_string(self, use_diff=False):
        try:
            return json.dumps(self.to_dict(), indent=2, default=str) + "\n"
        except Exception:  # pragma: no cover
            return str(self.to_dict())

    def __repr__(self):
        return f"{self.__class__.__name__} {self.to_json_string()}"

    # -- composition ----------------------------------------------------------------------------------
    def __add__(self, other):
        if isinstance(other, type(self)):
            for key, cfg in other.local_config.items():
                self.set_local(key, cfg)
            return self
        return ComposableConfig(configs=[self, other])

    def expand(self):
        """One config per element of the cartesian product of the list-valued tunable parameters."""
        tunable, fixed = {}, {}
        sig = inspect.signature(self.__init__).parameters
        for p in self.params_list:
            val = getattr(self, p)
            if val is None:
                continue
            default = sig[p].default if p in sig else None
            is_list_param = isinstance(default, (list, tuple, dict))
            if p not in self.non_tunable_params and isinstance(val, list) and not is_list_param:
                tunable[p] = val
            else:
                fixed[p] = val
        if not tunable:
            return [self]
        out = []
        names = list(tunable)
        for combo in itertools.product(*(tunable[n] for n in names)):
            cfg = self.__class__(**{**fixed, **dict(zip(names, combo))})
            cfg.local_config = dict(self.local_config)
            out.append(cfg)
        return out

    # -- model mapping ----------------------------------------------------------------------------------
    @staticmethod
    def _is_op_type(key):
        return not isinstance(key, str)

    @staticmethod
    def _op_type_to_str(op_type):
        return getattr(op_type, "__name__", "")

    def _get_op_name_op_type_config(self):
        by_type, by_name = {}, {}
        for key, cfg in self.local_config.items():
            if self._is_op_type(key):
                by_type[self._op_type_to_str(key)] = cfg
            else:
                by_name[key] = cfg
        return by_type, by_name

    def to_config_mapping(self, config_list=None, model_info=None):
        """(op_name, op_type) -> config, precedence global < op type < regex op name (reference :586-617)."""
        mapping = OrderedDict()
        for config in config_list or [self]:
            by_type, by_name = config._get_op_name_op_type_config()
            for op_name, op_type in model_info:
                if self.global_config is not None:
                    mapping[(op_name, op_type)] = config.global_config
                if op_type in by_type:
                    mapping[(op_name, op_type)] = by_type[op_type]
                for pattern, cfg in by_name.items():
                    if re.match(pattern, op_name):
                        mapping[(op_name, op_type)] = cfg
        return mapping

    @classmethod
    def register_supported_configs(cls):
        raise NotImplementedError

    @classmethod
    def get_config_set_for_tuning(cls):
        raise NotImplementedError

    @classmethod
    def validate(cls, user_config):
        return None

    def __eq__(self, other):
        if not isinstance(other, type(self)):
            return False
        same_params = self.params_list == other.params_list and all(
            getattr(self, str(a)) == getattr(other, str(a)) for a in self.params_list
        )
        return same_params and self.local_config == other.local_config and self.global_config == other.global_config

    __hash__ = None


class ComposableConfig(BaseConfig):
    """`RTNConfig() + GPTQConfig()`: several algorithm configs applied to one model."""

    name = "composable_config"

    def __init__(self, configs):
        object.__setattr__(self, "config_list", list(configs))

    def __setattr__(self, key, value):
        object.__setattr__(self, key, value)
        for cfg in self.config_list:
            if hasattr(cfg, key):
                setattr(cfg, key, value)

    def __add__(self, other):
        if isinstance(other, type(self)):
            self.config_list.extend(other.config_list)
        else:
            self.config_list.append(other)
        return self

    def to_dict(self):
        return {cfg.name: cfg.to_dict() for cfg in self.config_list}

    @classmethod
    def from_dict(cls, config_dict, config_registry):
        assert len(config_dict) >= 1, "The config dict must include at least one configuration."
        items = list(config_dict.items())
        config = config_registry[items[0][0]].from_dict(items[0][1])
        for algo, value in items[1:]:
            config = config + config_registry[algo].from_dict(value)
        return config

    def to_json_string(self, use_diff=False):
        return json.dumps(self.to_dict(), indent=2, default=str) + "\n"

    def to_config_mapping(self, config_list=None, model_info=None):
        mapping = OrderedDict()
        for cfg in self.config_list:
            info = model_info.get(cfg.name) if isinstance(model_info, dict) else model_info
            mapping.update(cfg.to_config_mapping(model_info=info))
        return mapping

    def get_model_info(self, model, *args, **kwargs):
        return {cfg.name: cfg.get_model_info(model, *args, **kwargs) for cfg in self.config_list}

    @classmethod
    def register_supported_configs(cls):
        raise NotImplementedError

    @classmethod
    def get_config_set_for_tuning(cls):
        return None
