from .factory import create_model_from_config, create_model_from_config_path

__all__ = ["create_model_from_config", "create_model_from_config_path"]
