"""Stand-in for PyTables (imported by the reference's driver script only)."""


def open_file(*a, **kw):
    raise NotImplementedError("tables stub")
