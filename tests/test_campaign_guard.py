"""The campaign's oracle-memory guard (tests/campaign.py::install_oracle_guard): a write into an oracle container's array while a
device-library call is under way is reported with the method, the array and the bytes; calls that leave the oracle alone report
nothing; what the test itself writes between two device calls is not a report."""
import types

import numpy as np

import campaign
import oracle_binding as ob
import parity


def test_a_write_behind_the_tests_back_is_reported(monkeypatch):
    class DeviceNetwork:                     # stands for snn_amd.DeviceNetwork: only the wrapping is under test
        def __init__(self):
            self.calls = 0

        def run(self, steps):
            self.calls += 1

        def poke(self, array, index, value):  # "the library" scribbling over host memory it does not own
            array.reshape(-1)[index] = value

        def _private(self):
            return 7

    fake = types.SimpleNamespace(DeviceNetwork=DeviceNetwork)
    monkeypatch.setattr(ob.Net, "__init__", ob.Net.__init__)          # (the guard replaces it: put back after the test)
    reports = []
    campaign.install_oracle_guard(fake, reports.append)
    net = parity.make_oracle(parity.Layout([(0, 4, 5)], [(3, 2, 2)]), st_kind=ob.ST_POISSON)
    dn = fake.DeviceNetwork()
    dn.run(5)
    net["current_voltage"] = np.float32(-60.0)                        # the test's own write, between two device calls
    dn.run(5)
    assert reports == [] and dn.calls == 2 and dn._private() == 7
    dn.poke(net["nt_flags"], 38, 0xFFFFFFFE)
    assert len(reports) == 1
    r = reports[0]
    assert r["oracle_memory_changed_during"] == "poke" and r["array"] == "nt_flags" and r["changed_bytes"] == 4 and r["first_byte"] == 38 * 4
    assert r["was"][:4] == [0, 0, 0, 0] and r["now"][:4] == [0xFE, 0xFF, 0xFF, 0xFF]
    net.run(3, voltage_history=True)
    dn.poke(net.voltage_history, 2, 1.0)                              # histories are guarded too
    assert len(reports) == 2 and reports[1]["array"] == "voltage_history"
