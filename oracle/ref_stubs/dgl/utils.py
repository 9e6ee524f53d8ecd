"""TEST INFRASTRUCTURE ONLY."""


def expand_as_pair(input_, g=None):
    if isinstance(input_, tuple):
        return input_
    return input_, input_
