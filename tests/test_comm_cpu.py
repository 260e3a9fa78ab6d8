"""File rendezvous of the RCCL unique id (eigd_amd/comm.py) -- host logic only, no GPU."""
import os
import threading
import time

import pytest

from eigd_amd import comm


def test_stale_id_of_an_earlier_launch_is_ignored(tmp_path, monkeypatch):
    monkeypatch.setenv("EIGD_COMM_DIR", str(tmp_path))
    stale = tmp_path / "uid1.bin"
    stale.write_bytes(b"OLD" * 10)
    old = time.time() - 3600.0
    os.utime(stale, (old, old))
    with pytest.raises(TimeoutError):
        comm.exchange_unique_id(1, 2, None, tag="uid1", timeout=0.3)
    got = {}
    t = threading.Thread(target=lambda: got.setdefault("uid", comm.exchange_unique_id(1, 2, None, tag="uid1", timeout=20)))
    t.start()
    time.sleep(0.2)
    assert comm.exchange_unique_id(0, 2, lambda: b"NEW" * 10, tag="uid1") == b"NEW" * 10   # rank 0 replaces the stale file
    t.join()
    assert got["uid"] == b"NEW" * 10
    comm.retire_unique_id(0, tag="uid1")
    assert not stale.exists()


def test_rendezvous_directory_must_be_private(tmp_path, monkeypatch):
    d = tmp_path / "shared"
    d.mkdir(mode=0o777)
    os.chmod(d, 0o777)
    monkeypatch.setenv("EIGD_COMM_DIR", str(d))
    with pytest.raises(PermissionError):
        comm.exchange_unique_id(0, 2, lambda: b"x", tag="uid")


def test_published_id_is_private_whatever_the_umask(tmp_path, monkeypatch):
    """under umask 002 (the default of non-root users on RHEL-family systems) a plain open() would publish the id 0664 and
    every other rank would reject it until its timeout: the writer sets the mode itself"""
    monkeypatch.setenv("EIGD_COMM_DIR", str(tmp_path / "rdv"))
    old = os.umask(0o002)
    try:
        assert comm.exchange_unique_id(0, 2, lambda: b"ID" * 16, tag="uid2") == b"ID" * 16
    finally:
        os.umask(old)
    st = os.stat(tmp_path / "rdv" / "uid2.bin")
    assert (st.st_mode & 0o777) == 0o600
    assert comm.exchange_unique_id(1, 2, None, tag="uid2", timeout=5.0) == b"ID" * 16
