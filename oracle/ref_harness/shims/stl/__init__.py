class mesh:
    pass
