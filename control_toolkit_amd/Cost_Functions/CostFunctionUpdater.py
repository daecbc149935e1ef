"""CostFunctionUpdater — mirror of reference Cost_Functions/CostFunctionUpdater.py:9-68: watches the cost
YAML; on modification re-reads `[environment_name][cost_function_name]` into `cost_function.config` and
raises `reload_cost_parameters_from_config_flag`, which `controller_mpc.step` consumes at its top
(Controllers/controller_mpc.py:101 -> cost_function_wrapper.py:71-74).

The reference watches with the `watchdog` package (an Observer thread).  That package is not a dependency of
this build: the watcher here is a daemon thread that polls the file's (mtime_ns, size) — same contract (one
watcher per path, `stop`, `stop_all_watchers`, the flag is only ever SET from the thread), plus `poll_now()`
so that a caller (or a test) can force the check deterministically."""
import atexit
import os
import threading

from yaml import safe_load


class CostFunctionUpdater:
    active_watchers = {}          # path -> updater (reference :11)
    POLL_SECONDS = 0.2

    def __init__(self, cost_function, environment_name, cost_function_name, start_thread: bool = True):
        self.cost_function = cost_function
        self.environment_name = environment_name
        self.cost_function_name = cost_function_name
        self._stop = threading.Event()
        self._thread = None
        if not hasattr(cost_function, "config_path") or cost_function.config_path is None:
            return                                                      # reference :15: nothing to watch
        self.config_path = os.path.abspath(cost_function.config_path)
        if not os.path.isfile(self.config_path):
            raise FileNotFoundError(f"Configuration file not found at path: {self.config_path}")   # reference :18-19
        if self.config_path in CostFunctionUpdater.active_watchers:     # reference :22-24
            CostFunctionUpdater.active_watchers[self.config_path].stop()
        self._stamp = self._stat()
        CostFunctionUpdater.active_watchers[self.config_path] = self
        if start_thread:
            self._thread = threading.Thread(target=self._run, name="CostFunctionUpdater", daemon=True)
            self._thread.start()
            atexit.register(self.stop)

    def _stat(self):
        st = os.stat(self.config_path)
        return (st.st_mtime_ns, st.st_size)

    def poll_now(self) -> bool:
        """One check; True if the file changed and the flag was raised (reference on_modified, :63-66)."""
        if not hasattr(self, "config_path"):
            return False
        try:
            stamp = self._stat()
        except OSError:
            return False                                                # mid-rename by an editor: next poll sees it
        if stamp == self._stamp:
            return False
        try:
            section = safe_load(open(self.config_path, "r"))[self.environment_name][self.cost_function_name]
        except Exception:
            return False                                                # half-written file: keep the old config, retry
        self._stamp = stamp
        # validate HERE, not in the control loop: an invalid edit (typo, unknown key) keeps the config in force and is
        # logged, so controller_mpc.step never raises because of a file somebody is editing
        check = getattr(self.cost_function, "validate_config", None)
        if check is not None:
            try:
                check(section)
            except ValueError as e:
                import logging
                logging.getLogger(__name__).warning("%s: edit ignored, previous cost parameters stay in force (%s)", self.config_path, e)
                return False
        self.cost_function.config = section
        self.cost_function.reload_cost_parameters_from_config_flag = True
        return True

    def _run(self):
        while not self._stop.wait(self.POLL_SECONDS):
            self.poll_now()

    def stop(self):
        self._stop.set()
        t, self._thread = self._thread, None
        if t is not None and t.is_alive() and t is not threading.current_thread():
            t.join(timeout=2.0)
        path = getattr(self, "config_path", None)
        if path is not None and CostFunctionUpdater.active_watchers.get(path) is self:
            del CostFunctionUpdater.active_watchers[path]

    def __del__(self):
        try:
            self.stop()
        except Exception:
            pass

    @classmethod
    def stop_all_watchers(cls):
        for w in list(cls.active_watchers.values()):
            w.stop()
        cls.active_watchers.clear()
