class VariableParameters:
    def __init__(self, lib):
        self.lib = lib

    def set_attributes(self, attributes, device=None):
        for k, v in attributes.items():
            setattr(self, k, v)

    def update_attributes(self, attributes):
        for k, v in attributes.items():
            setattr(self, k, v)
