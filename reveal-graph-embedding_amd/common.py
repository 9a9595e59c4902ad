"""Mirror of the two helpers of reveal_graph_embedding/common.py the ARCTE entry point uses
(reference common.py:24-33, 36-49)."""
import multiprocessing


def get_threads_number():
    """Number of parallel tasks the entry point asks for when -nt is not given (reference: the CPU count,
    falling back to 8).  arcte() clamps it to the number of visible GPUs."""
    try:
        return multiprocessing.cpu_count()
    except NotImplementedError:
        return 8


def get_file_row_generator(file_path, separator, encoding=None):
    """Yields the separator-split fields of every line of a text file."""
    with open(file_path, encoding=encoding) as file_object:
        for line in file_object:
            yield line.strip().split(separator)
