from .mustache import main

main()
