"""ORACLE / TEST INFRASTRUCTURE ONLY. Stub for pyzmq: the reference imports it in a dev tool
(graph_ltpl/testing_tools/src/objectlist_dummy.py:2) whose sockets are only opened under __main__."""
PUB = 1
SUB = 2
SNDMORE = 2


class Context(object):
    def socket(self, *args, **kwargs):
        raise RuntimeError("zmq is stubbed in the oracle environment")

    def term(self):
        pass
