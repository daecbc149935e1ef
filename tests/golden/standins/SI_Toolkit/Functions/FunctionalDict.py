class FunctionalDict(dict):
    pass
