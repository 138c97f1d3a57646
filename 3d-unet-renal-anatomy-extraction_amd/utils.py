"""Drop-in `utils` module (reference utils.py:5-13): the two JSON helpers data.py imports."""
import json


def json_save(path, data):
    with open(path, "w") as f:
        json.dump(data, f, indent=2)


def json_load(path):
    with open(path, "r") as f:
        return json.load(f)
