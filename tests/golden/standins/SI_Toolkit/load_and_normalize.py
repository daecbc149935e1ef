import yaml


def load_yaml(path, mode="r"):
    with open(path, mode) as f:
        return yaml.safe_load(f)
