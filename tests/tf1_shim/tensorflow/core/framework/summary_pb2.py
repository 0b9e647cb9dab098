"""stand-in for the two protobuf messages the reference builds by hand (rbm/base_rbm.py:584-589, dbm.py:819-823)"""


class Summary(object):
    class Value(object):
        def __init__(self, tag=None, simple_value=None):
            self.tag, self.simple_value = tag, simple_value

    def __init__(self, value=None):
        self.value = list(value or [])
