"""Run tests/test_gpu_dp.py::_nccl_worker in-process (stderr visible)."""
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class Q:
    def put(self, v):
        print(v)


if __name__ == '__main__':
    from tests.test_gpu_dp import _nccl_worker
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    _nccl_worker((Q(), port))
