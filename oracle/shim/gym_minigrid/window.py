"""gym_minigrid.window stub: interactive matplotlib window is only used by the
reference's manual tools (scripts/manual_control.py) -- out of scope."""


class Window(object):
    def __init__(self, title):
        raise RuntimeError("interactive Window is not available in the oracle shim")
