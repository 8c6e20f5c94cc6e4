from .modeling_auto import AutoModel, AutoModelForCausalLM, AutoModelForSeq2SeqLM, _BaseINCAutoModelClass

__all__ = ["AutoModel", "AutoModelForCausalLM", "AutoModelForSeq2SeqLM", "_BaseINCAutoModelClass"]
