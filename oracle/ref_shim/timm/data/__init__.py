def resolve_model_data_config(*a, **k): raise RuntimeError('unused')
def resolve_data_config(*a, **k): raise RuntimeError('unused')
