def get_thread_id():
    return 0


_get_thread_id = get_thread_id
