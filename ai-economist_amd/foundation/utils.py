"""Episode-log files (reference: F/utils.py:18-43): the dense log of an environment's last logged
episode as JSON inside one lz4 frame -- the format tutorials/utils/plotting.py reads.

lz4 is an optional dependency, as in the reference (`pip install lz4`); without it these two functions
raise ImportError and nothing else in the package is affected."""
import json


def save_episode_log(game_object, filepath, compression_level=16):
    """Writes game_object.previous_episode_dense_log to `filepath`; compression_level is clamped to 0..16."""
    import lz4.frame

    log = game_object.previous_episode_dense_log
    level = min(16, max(0, int(compression_level)))
    payload = json.dumps(log, ensure_ascii=False).encode("utf-8")
    with lz4.frame.open(filepath, mode="wb", compression_level=level) as out:
        out.write(payload)


def load_episode_log(filepath):
    """Reads a log written by save_episode_log (this package's or the reference's)."""
    import lz4.frame

    with lz4.frame.open(filepath, mode="rb") as src:
        return json.loads(src.read())
