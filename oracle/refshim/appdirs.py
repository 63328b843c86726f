"""Stand-in for `appdirs` (basis-cache location only; off the hot path)."""
import tempfile


def user_data_dir(appname=None, appauthor=None, *a, **k):
    return tempfile.gettempdir()
