"""Second half of make_ref_vectors.py: acquisition, code generators, settings, preRun (filled in below)."""


def gen_acq(only=None):
    print("[acq] not implemented yet")


def gen_codes(only=None):
    print("[codes] not implemented yet")


def gen_settings(only=None):
    print("[settings] not implemented yet")


def gen_prerun(only=None):
    print("[prerun] not implemented yet")
