class Observer:  # never started: the stand-in cost function has no config_path
    def schedule(self, *a, **k): return None
    def start(self): pass
    def stop(self): pass
    def join(self): pass
    def is_alive(self): return False
